"""CPU-side checks: the C-ABI library loads and exports every declared symbol, fails loudly without
a device, and the host-side formats agree with the oracle.  No compute kernels run here."""
import ctypes as C
import os
import re
import struct
import subprocess

import numpy as np
import pytest

from conftest import ROOT, has_gpu


@pytest.fixture(scope="module")
def hg():
    import hypergen_amd
    hypergen_amd.lib()
    return hypergen_amd


def test_every_declared_symbol_is_exported(hg):
    hdr = open(os.path.join(ROOT, "include", "hypergen.h")).read()
    hdr = re.sub(r"/\*.*?\*/", "", hdr, flags=re.S)  # declarations only, not prose
    declared = set(re.findall(r"\b(hg_[a-z0-9_]+)\s*\(", hdr))
    assert len(declared) >= 30
    L = hg.lib()
    for name in sorted(declared):
        assert hasattr(L, name), name
    assert declared == set(hg.EXPORTS)
    nm = subprocess.run(["nm", "-D", "--defined-only", hg.LIB_PATH], capture_output=True, text=True).stdout
    assert not re.search(r"\borc_", nm), "the product must not contain the oracle"


def test_kernel_names_in_the_header_exist_in_the_library(hg):
    """the header's prose names kernels (hg_ctx_last_kernel's examples, the ksize ranges, the Hamming paths): every such
    name must be a kernel the built library really contains -- prose that drifts from the code fails here"""
    hdr = open(os.path.join(ROOT, "include", "hypergen.h")).read()
    names = set(re.findall(r"\b(kmer_sample_[a-z0-9_]+|[a-z0-9_]+_kernel)\b(?!\.cu)", hdr)) - {"cuda_kernel"}
    names = {n for n in names if not n.startswith("hg_")}  # (hg_ctx_last_kernel is an entry point, not a kernel)
    assert {"kmer_sample_shared", "kmer_sample_long", "dist_mfma_kernel"} <= names
    nm = subprocess.run(["nm", "-C", hg.LIB_PATH], capture_output=True, text=True).stdout
    for n in sorted(names):
        assert re.search(r"::%s[<(]" % re.escape(n), nm), "include/hypergen.h names `%s`, the library has no such kernel" % n
    # ... and fully spelled instantiations (name<args>) must exist with exactly those template arguments
    for full in re.findall(r"\"((?:kmer_sample|dist_mfma)[a-z_]*<[^\">]+>)\"", hdr):
        assert ("::" + full + "(") in nm, "include/hypergen.h spells `%s`, no such instantiation in the library" % full


def test_product_does_not_reference_oracle():
    for dp, _, fs in os.walk(os.path.join(ROOT, "hyper-gen_amd")):
        for f in fs:
            if f.endswith((".hip", ".cpp", ".h", ".py")):
                txt = open(os.path.join(dp, f), errors="ignore").read()
                assert "hg_oracle" not in txt and "from oracle" not in txt and "import oracle" not in txt, f


@pytest.mark.skipif(has_gpu(), reason="only meaningful on a box without a GPU")
def test_fails_loudly_without_device(hg):
    with pytest.raises(hg.HgError) as e:
        hg.Context(0)
    assert e.value.status == hg.ERR_NO_DEVICE


def test_status_strings_and_defaults(hg):
    L = hg.lib()
    assert L.hg_status_str(0) == b"ok" and b"capacity" in L.hg_status_str(hg.ERR_CAPACITY)
    p = hg.default_params()
    assert (p.ksize, p.scaled, p.seed, p.canonical, p.hv_d) == (21, 1500, 123, 1, 4096)  # src/types.rs:97-113


def test_pack_matches_oracle(hg, orc):
    rng = np.random.default_rng(0)
    for amp, d in ((0, 256), (31, 4096), (32, 4096), (300, 4096), (1000, 1024), (20000, 512)):
        hv = rng.integers(-amp, amp + 1, d).astype(np.int16)
        q, packed = hg.hv_pack(hv)
        oq, opacked = orc.pack_hv(hv)
        assert q == oq == hg.hv_quant_bits(hv) and (packed == opacked).all()
        if q < 16:  # q == 16 reproduces the reference's sign-extension bug (src/hd.rs:140-141) bit for bit
            assert (hg.hv_unpack(packed, d, q) == hv).all()
        else:
            assert (hg.hv_unpack(packed, d, q) == orc.unpack_hv(opacked, d, q)).all()
    # every width incl. the reference's lossy q = 16 (src/hd.rs:140-141), and dimensions that are not whole
    # 256-blocks (the reference packs hv_d / 256 blocks into q * (hv_d >> 3) bytes and loses the rest, src/hd.rs:143-147).
    # The oracle is written from the crate's format definition (scatter by element), the product follows the crate's
    # accumulator loop; tests/test_oracle.py checks the oracle against a third, bit-level model.
    for q in range(6, 17):
        for d in (256, 100, 264, 1000, 4104):
            amp = (1 << (q - 1)) - 1
            hv = rng.integers(-amp - 1, amp + 1, d).astype(np.int16)
            _, packed = hg.hv_pack(hv, q)
            _, opacked = orc.pack_hv(hv, q)
            assert packed.size == opacked.size == q * (d >> 3) == hg.lib().hg_hv_packed_bytes(d, q)
            assert (packed == opacked).all(), (q, d)
            assert (hg.hv_unpack(packed, d, q) == orc.unpack_hv(opacked, d, q)).all(), (q, d)


def test_naive_payload_layout_matches_oracle_and_is_told_apart(hg, orc):
    """the non-AVX2 payload layout (src/hd.rs:158-166, 213-231): product == oracle byte for byte and value for value (the
    reference's lossy corners included), and the two layouts are told apart by their lengths"""
    rng = np.random.default_rng(21)
    for q in range(6, 17):
        for d in (256, 1000, 4096, 4104, 24):
            lim = 1 << (q - 1)
            hv = rng.integers(-lim, lim, d).astype(np.int16)
            hv[:3] = [-lim, lim - 1, 0]
            _, got = hg.hv_pack_naive(hv, q)
            _, want = orc.pack_hv_naive(hv, q)
            assert got.size == hg.lib().hg_hv_packed_bytes_naive(d, q) and np.array_equal(got, want.view(np.uint8)), (q, d)
            assert np.array_equal(hg.hv_unpack_naive(got, d, q), orc.unpack_hv_naive(want, d, q)), (q, d)
            bp = hg.lib().hg_hv_packed_bytes(d, q) // 2 * 2  # (the i16 view of the BitPacker8x bytes)
            assert hg.hv_payload_layout(d, q, got.size) == hg.PAYLOAD_NAIVE
            assert hg.hv_payload_layout(d, q, bp) == hg.PAYLOAD_BITPACKER8X
            assert hg.hv_payload_layout(d, q, bp + 1) == -1 and hg.hv_payload_layout(d, 0, bp) == -1
    # a value wider than q bits is cut to its low q bits, like the reference's `(hv >> k) & 1` loop
    hv = np.array([300, -300] * 8, np.int16)
    assert np.array_equal(hg.hv_pack_naive(hv, 6)[1], orc.pack_hv_naive(hv, 6)[1].view(np.uint8))


def test_sketch_file_image_reader(hg, tmp_path):
    """hg_sketch_file_read_image: the same records without payload copies; payload i sits at its offset in the image"""
    rng = np.random.default_rng(22)
    recs = []
    for i in range(7):
        hv = rng.integers(-200, 200, 512).astype(np.int16)
        q, pk = (hg.hv_pack_naive if i == 3 else hg.hv_pack)(hv)
        recs.append(dict(ksize=21, scaled=1500, canonical=True, seed=123, hv_d=512, hv_quant_bits=q, hv_norm_2=i,
                         file_str="p" * i + "/g%d.fna" % i, hv=pk.view(np.int16)))
    path = str(tmp_path / "x.sketch")
    hg.write_sketch_file(path, recs)
    img, meta = hg.read_sketch_file_image(path)
    full = hg.read_sketch_file(path)
    assert img.tobytes() == open(path, "rb").read() and len(meta) == len(full) == 7
    for m, f, r in zip(meta, full, recs):
        assert m["file_str"] == f["file_str"] and m["hv_norm_2"] == f["hv_norm_2"] and m["payload_bytes"] == f["hv"].size * 2
        assert img[m["payload_off"]: m["payload_off"] + m["payload_bytes"]].tobytes() == f["hv"].tobytes() == r["hv"].tobytes()
    assert hg.hv_payload_layout(512, meta[3]["hv_quant_bits"], meta[3]["payload_bytes"]) == hg.PAYLOAD_NAIVE
    assert hg.hv_payload_layout(512, meta[2]["hv_quant_bits"], meta[2]["payload_bytes"]) == hg.PAYLOAD_BITPACKER8X


def test_sketch_file_layout_and_roundtrip(hg, tmp_path):
    hv = (np.arange(1536) - 700).astype(np.int16)
    recs = [dict(ksize=21, scaled=1500, canonical=True, seed=123, hv_d=4096, hv_quant_bits=6, hv_norm_2=-5,
                 file_str="dir/a.fna", hv=hv),
            dict(ksize=31, scaled=7, canonical=False, seed=2**63 + 5, hv_d=256, hv_quant_bits=16, hv_norm_2=2**31 - 1,
                 file_str="b.fa", hv=np.zeros(0, np.int16))]
    path = str(tmp_path / "x.sketch")
    hg.write_sketch_file(path, recs)
    raw = open(path, "rb").read()
    # bincode 1.x defaults (src/types.rs:224-235 field order): u64 count, then per record
    # u8 ksize, u64 scaled, u8 canonical, u64 seed, u64 hv_d, u8 quant_bits, i32 norm, str, Vec<i16>
    assert struct.unpack_from("<Q", raw, 0)[0] == 2
    assert struct.unpack_from("<BQBQQBi", raw, 8) == (21, 1500, 1, 123, 4096, 6, -5)
    off = 8 + 31
    assert struct.unpack_from("<Q", raw, off)[0] == 9 and raw[off + 8:off + 17] == b"dir/a.fna"
    off += 17
    assert struct.unpack_from("<Q", raw, off)[0] == 1536
    assert np.frombuffer(raw, "<i2", 1536, off + 8).tolist() == hv.tolist()
    back = hg.read_sketch_file(path)
    for a, b in zip(recs, back):
        for k in a:
            assert (np.array_equal(a[k], b[k]) if k == "hv" else a[k] == b[k]), k
    open(path, "wb").write(raw[:50])
    with pytest.raises(hg.HgError):
        hg.read_sketch_file(path)


def test_read_merge_seq(hg, orc, tmp_path):
    txt = b">r1 x\nACGTNN\nacgu\r\n>r2\n\nTT\r"
    p = tmp_path / "t.fna"
    p.write_bytes(txt)
    got = hg.read_merge_seq(str(p))
    assert bytes(got) == b"NACGTNNacguNTT" and (got == orc.read_merge_seq(txt)).all()
    import gzip
    gz = tmp_path / "t2.fna"  # gzip content behind a .fna name: sniffed like needletail does
    gz.write_bytes(gzip.compress(txt))
    assert bytes(hg.read_merge_seq(str(gz))) == b"NACGTNNacguNTT"
    big = b">x\n" + b"ACGT" * 3_000_000 + b"\n>y\n" + b"TTGA" * 10
    gz.write_bytes(gzip.compress(big))
    got = hg.read_merge_seq(str(gz))
    assert got.size == 2 + 12_000_000 + 40 and (got == orc.read_merge_seq(big)).all()
    ref_fixture = b">test_seq\nAGCTCTTANNAGCCCNTTacgttacagccctgaaaacttt"
    p.write_bytes(ref_fixture)
    assert bytes(hg.read_merge_seq(str(p))) == b"NAGCTCTTANNAGCCCNTTacgttacagccctgaaaacttt"


def test_read_needletail_mode_fasta_blanks_and_fastq(hg, orc, tmp_path):
    """-D cpu reads files the way needletail does (src/sketch.rs:76-87): blanks inside sequence lines vanish,
    FASTQ records contribute their sequence line only; -D gpu keeps read_merge_seq's line semantics."""
    import gzip
    fa = b">r1 desc\nAC GT\tAC\r\nacgu \r\n\n>r2\nTT N\r"
    p = tmp_path / "t.fa"
    p.write_bytes(fa)
    got = hg.read_merge_seq(str(p), hg.READ_NEEDLETAIL)
    assert bytes(got) == b"NACGTACacguNTTN" and (got == orc.read_needletail(fa)).all()
    assert bytes(hg.read_merge_seq(str(p))) == b"NAC GT\tACacgu NTT N"  # reference GPU reader: lines as they are
    fq = b"@read1 x\nACGTTGCA\n+\nIIIIIIII\n@read2\nGGGACCC\r\n+read2\n>>>>III\n"
    p.write_bytes(fq)
    got = hg.read_merge_seq(str(p), hg.READ_NEEDLETAIL)
    assert bytes(got) == b"NACGTTGCANGGGACCC" and (got == orc.read_needletail(fq)).all()
    p.write_bytes(gzip.compress(fq))  # a quality line may start with '>' or '@': the 4-line cadence decides
    assert bytes(hg.read_merge_seq(str(p), hg.READ_NEEDLETAIL)) == b"NACGTTGCANGGGACCC"
    rng = np.random.default_rng(5)
    for trial in range(20):  # random mixes of record lengths, blanks, line ends
        recs = []
        for r in range(rng.integers(1, 5)):
            seq = bytes(rng.choice(list(b"ACGTacgtNn \t"), rng.integers(0, 200)).astype(np.uint8))
            lines = [seq[i:i + 37] for i in range(0, len(seq), 37)]
            recs.append(b">h%d\n" % r + (b"\r\n" if trial % 2 else b"\n").join(lines))
        txt = b"\n".join(recs) + (b"\n" if trial % 3 else b"")
        p.write_bytes(txt)
        assert (hg.read_merge_seq(str(p), hg.READ_NEEDLETAIL) == orc.read_needletail(txt)).all(), trial


def test_corrupt_sketch_files_are_rejected_not_fatal(hg, tmp_path):
    path = str(tmp_path / "bad.sketch")
    for count in (2**63, 2**40, 3):  # record counts the file cannot hold
        open(path, "wb").write(struct.pack("<Q", count) + b"\x00" * 60)
        with pytest.raises(hg.HgError):
            hg.read_sketch_file(path)
    hdr = struct.pack("<BQBQQBi", 21, 1500, 1, 123, 4096, 6, 0)
    open(path, "wb").write(struct.pack("<Q", 1) + hdr + struct.pack("<Q", 2**62) + b"ab")  # absurd path length
    with pytest.raises(hg.HgError):
        hg.read_sketch_file(path)
    open(path, "wb").write(struct.pack("<Q", 1) + hdr + struct.pack("<Q", 1) + b"a" + struct.pack("<Q", 2**61))
    with pytest.raises(hg.HgError):
        hg.read_sketch_file(path)


def test_sort_ani_hits_matches_dump_ani_file_order(hg):
    # model of src/utils.rs:262-269 on the row-major enumeration of src/dist.rs:251-265
    rng = np.random.default_rng(3)
    R, Q = 7, 9
    ani = rng.choice([90.0, 95.5, 99.0, 100.0], (R, Q)).astype(np.float32)
    pairs = [(i, j) for i in range(R) for j in range(Q)]
    idx = sorted(range(len(pairs)), key=lambda t: ani[pairs[t]])  # stable ascending
    idx.reverse()
    want = [pairs[t] for t in idx]
    hits = np.zeros(R * Q, hg.ANI_HIT_DTYPE)
    perm = rng.permutation(R * Q)
    for o, t in enumerate(perm):
        hits[o] = (pairs[t][0], pairs[t][1], ani[pairs[t]])
    got = hg.sort_ani_hits(hits, Q)
    assert [(int(h["ref_idx"]), int(h["qry_idx"])) for h in got] == want


def test_cli_surface(hg):
    cli = hg.CLI_PATH
    assert os.path.exists(cli)
    out = subprocess.run([cli, "--help"], capture_output=True, text=True)
    assert out.returncode == 0 and "sketch" in out.stdout and "--ani_th" in out.stdout
    assert subprocess.run([cli, "search"], capture_output=True).returncode == 0  # no-op like src/main.rs:22-24
    bad = subprocess.run([cli, "sketch", "-p", "/tmp", "-o", "/tmp/x", "-t", "300"], capture_output=True, text=True)
    assert bad.returncode != 0 and "invalid value" in bad.stderr  # -t is u8 (src/utils.rs:54-56)
    assert subprocess.run([cli, "frobnicate"], capture_output=True).returncode != 0
    # the extension flags: documented, value-checked before any device is touched
    assert "--pack_layout" in out.stdout and "--shards" in out.stdout and "--top_n" in out.stdout
    bad = subprocess.run([cli, "sketch", "-p", "/tmp", "-o", "/tmp/x", "--pack_layout", "zip"], capture_output=True, text=True)
    assert bad.returncode != 0 and "pack_layout" in bad.stderr
    bad = subprocess.run([cli, "dist", "-r", "/nonexistent", "-q", "/nonexistent", "-o", "/tmp/x", "--shards", "65"], capture_output=True, text=True)
    assert bad.returncode != 0 and "invalid value" in bad.stderr


def test_reader_blocks_equal_whole_file_semantics(hg, orc, tmp_path):
    """The reader works through a file in 256 KiB blocks of whole lines: line ends, CRs, headers, FASTQ records and a
    line longer than a block must come out as if the file had been merged in one piece (the oracle's whole-buffer
    restatement), in both read modes, also 2-bit packed, also gzip-ed."""
    import gzip
    rng = np.random.default_rng(77)
    alpha = np.frombuffer(b"ACGTacgtNnU", np.uint8)

    def seq(n):
        return rng.choice(alpha, n, p=[.22, .22, .22, .22, .02, .02, .02, .02, .02, .01, .01]).tobytes()

    def fasta(total, width, crlf=False, blanks=False):
        out, left = [], total
        while left > 0:
            out.append(b">rec %d\n" % len(out))
            m = min(left, int(rng.integers(1, 400_000)))
            body = seq(m)
            for j in range(0, m, width):
                line = body[j:j + width]
                if blanks and rng.random() < 0.01 and len(line) > 4:
                    line = line[:2] + b" \t" + line[2:]
                out.append(line + (b"\r\n" if crlf else b"\n"))
            left -= m
        return b"".join(out)

    def fastq(records):
        out = []
        for r in range(records):
            m = int(rng.integers(20, 3000))
            out.append(b"@r%d\n" % r + seq(m) + b"\n+\n" + bytes(rng.integers(33, 74, m, dtype=np.uint8)) + b"\n")
        return b"".join(out)

    cases = {
        "w80": fasta(1_500_000, 80), "w61_crlf": fasta(900_000, 61, crlf=True), "blanks": fasta(700_000, 70, blanks=True),
        "one_long_line": b">x\n" + seq(1_200_000) + b"\n>y\n" + seq(300_000),  # no trailing newline either
        "exact_block": b">a\n" + seq((256 << 10) - 4) + b"\n" + seq(5000) + b"\n",  # a '\\n' as the block's last byte
        "fastq": fastq(900), "tiny": b">t\nACGT", "empty": b"",
    }
    for name, txt in cases.items():
        for gz in (False, True):
            f = tmp_path / ("%s%s.fna" % (name, "_gz" if gz else ""))
            f.write_bytes(gzip.compress(txt, 1) if gz else txt)
            for mode, want, norm in ((hg.READ_MERGE, orc.read_merge_seq(txt), 0), (hg.READ_NEEDLETAIL, orc.read_needletail(txt), 1)):
                got = hg.read_merge_seq(str(f), mode)
                assert got.size == want.size and (got == want).all(), (name, gz, mode)
                p, n, cap = C.c_void_p(), C.c_size_t(0), C.c_size_t(0)
                flags = mode | 16 | (32 if norm else 0)
                assert hg.lib().hg_read_fastx_into(str(f).encode(), flags, C.byref(p), C.byref(cap), C.byref(n)) == 0
                try:
                    assert n.value == want.size
                    size = hg.lib().hg_pack2_size(n.value)
                    blob = np.ctypeslib.as_array(C.cast(p, C.POINTER(C.c_uint8)), shape=(max(size, 1),))[:size]
                    assert (blob == hg.pack2(want, norm)).all(), (name, gz, mode, "packed")
                finally:
                    hg.lib().hg_free(p)
