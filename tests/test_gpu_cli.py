"""End-to-end drop-in surface on the GPU: `hyper-gen sketch` on a directory of FASTA files, the
.sketch it writes, `hyper-gen dist`, and the ANI TSV -- against a model assembled from the oracle
(src/sketch.rs:12-69, src/dist.rs:11-63, src/utils.rs:260-308)."""
import os
import subprocess

import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def write_fasta(path, seq, name, width=80, two_records=False):
    s = bytes(seq).decode()
    with open(path, "w") as f:
        if two_records:
            half = len(s) // 2
            parts = [(name + "_a", s[:half]), (name + "_b extra words", s[half:])]
        else:
            parts = [(name, s)]
        for n, body in parts:
            f.write(">%s\n" % n)
            for i in range(0, len(body), width):
                f.write(body[i:i + width] + "\n")


def model_tsv(orc, hg, files, merged, params, ani_th, sym_with=None):
    recs = []
    for f, m in zip(files, merged):
        hv, n2, nh = orc.sketch_genome(m, ksize=params["k"], scaled=params["s"], norm=orc.NORM_U2T)
        recs.append((f, hv, n2))
    hvs = np.stack([r[1] for r in recs])
    n2s = np.array([r[2] for r in recs], np.int32)
    ani = orc.ani_matrix(hvs, n2s, hvs, n2s, params["k"])
    pairs = [(i, j) for i in range(len(files)) for j in range(i + 1, len(files))]  # symmetric: i < j
    order = sorted(range(len(pairs)), key=lambda t: ani[pairs[t]])
    order.reverse()
    lines = []
    for t in order:
        i, j = pairs[t]
        if ani[i, j] >= ani_th:
            lines.append((files[i], files[j], float(ani[i, j])))
    return recs, lines


def test_cli_sketch_dist_roundtrip(tmp_path, orc):
    import hypergen_amd as hg
    d = tmp_path / "fa"
    d.mkdir()
    L = 300_000
    gen = {"g000.fna": 0, "g010.fna": 10, "g030.fna": 30, "g100.fa": 100, "g101.fasta": 101}
    for name, g in gen.items():
        seq = orc.synth_genome(g, L)[1:]
        write_fasta(str(d / name), seq, name, two_records=(g == 10))
    files = sorted(str(d / n) for n in gen if n.endswith(".fna")) + [str(d / "g100.fa"), str(d / "g101.fasta")]
    merged = [hg.read_merge_seq(f) for f in files]
    out = str(tmp_path / "all.sketch")
    r = subprocess.run([hg.CLI_PATH, "sketch", "-p", str(d), "-o", out, "-s", "100", "-t", "4"],
                       capture_output=True, text=True)
    assert r.returncode == 0, r.stderr
    assert "Sketching 5 files took" in r.stdout and "Dump sketch file to" in r.stdout

    recs_model, lines_model = model_tsv(orc, hg, files, merged, {"k": 21, "s": 100}, 85.0)
    recs = hg.read_sketch_file(out)
    assert [x["file_str"] for x in recs] == files  # glob order: *.fna, *.fa, *.fasta (src/utils.rs:208-221)
    for x, (f, hv, n2) in zip(recs, recs_model):
        assert (x["ksize"], x["scaled"], x["seed"], x["canonical"], x["hv_d"]) == (21, 100, 123, True, 4096)
        assert x["hv_norm_2"] == n2
        q, packed = orc.pack_hv(hv)
        assert x["hv_quant_bits"] == q and (x["hv"].view(np.uint8) == packed).all()
        assert (hg.hv_unpack(x["hv"].view(np.uint8), 4096, q) == hv).all()

    tsv = str(tmp_path / "ani.tsv")
    r = subprocess.run([hg.CLI_PATH, "dist", "-r", out, "-q", out, "-o", tsv, "-a", "85"], capture_output=True, text=True)
    assert r.returncode == 0, r.stderr
    got = [l.split("\t") for l in open(tsv).read().splitlines()]
    assert len(got) == len(lines_model) >= 3
    for (a, b, v), (ma, mb, mv) in zip(got, lines_model):
        assert (a, b) == (ma, mb)
        assert v == "%.3f" % mv  # the oracle's float32 value, printed like "{:.3}": the same text
    # ref != query path: all R x Q pairs, including self pairs at 100.000
    out2 = str(tmp_path / "copy.sketch")
    hg.write_sketch_file(out2, recs)
    tsv2 = str(tmp_path / "ani2.tsv")
    r = subprocess.run([hg.CLI_PATH, "dist", "-r", out, "-q", out2, "-o", tsv2, "-a", "99.5"], capture_output=True, text=True)
    assert r.returncode == 0, r.stderr
    rows = [l.split("\t") for l in open(tsv2).read().splitlines()]
    assert len(rows) >= 5 and all(x[2] == "100.000" for x in rows[:5])


@pytest.mark.parametrize("scaled", [1, 5, 20])
def test_cli_sketch_dense_sampling(tmp_path, orc, scaled):
    """`hyper-gen sketch -s 1 | 5 | 20`: every k-mer, or one in 5 / 20, through the streaming path -- hit lists sized to the
    rate, hash sets of 20 000 .. 400 000 keys through the value-bucketed sort, HVs whose counts need 11+ bits -- against the
    oracle's sketch of the same files."""
    import hypergen_amd as hg
    d = tmp_path / "fa"
    d.mkdir()
    L = 400_000
    files = []
    for g in (3, 4, 5, 6):
        seq = orc.synth_genome(g, L + 1000 * g)[1:].copy()
        if g == 4:
            seq[20_000:20_000 + 171 * 150] = np.tile(seq[500:671].copy(), 150)  # a tandem repeat
        write_fasta(str(d / ("g%03d.fna" % g)), seq, "g%d" % g, two_records=(g == 5))
        files.append(str(d / ("g%03d.fna" % g)))
    out = str(tmp_path / "dense.sketch")
    r = subprocess.run([hg.CLI_PATH, "sketch", "-p", str(d), "-o", out, "-s", str(scaled), "-t", "4"], capture_output=True, text=True)
    assert r.returncode == 0, r.stderr
    recs = hg.read_sketch_file(out)
    assert [x["file_str"] for x in recs] == sorted(files)
    for x in recs:
        hv, n2, nh = orc.sketch_genome(hg.read_merge_seq(x["file_str"]), scaled=scaled, norm=orc.NORM_U2T)
        assert nh > 8192 and x["scaled"] == scaled and x["hv_norm_2"] == n2
        q, packed = orc.pack_hv(hv)
        assert x["hv_quant_bits"] == q and (x["hv"].view(np.uint8) == packed).all(), (scaled, x["file_str"])


def test_cli_reference_fixture(tmp_path):
    """BASELINE config 1: the reference's test/test.fna (40 bases) -> empty hash set, zero HV, 6-bit payload."""
    import hypergen_amd as hg
    d = tmp_path / "t"
    d.mkdir()
    (d / "test.fna").write_text(">test_seq\nAGCTCTTANNAGCCCNTTacgttacagccctgaaaacttt")
    out = str(tmp_path / "t.sketch")
    r = subprocess.run([hg.CLI_PATH, "sketch", "-p", str(d), "-o", out], capture_output=True, text=True)
    assert r.returncode == 0, r.stderr
    rec = hg.read_sketch_file(out)
    assert len(rec) == 1 and rec[0]["hv_norm_2"] == 0 and rec[0]["hv_quant_bits"] == 6 and rec[0]["hv"].size == 1536
    assert (hg.hv_unpack(rec[0]["hv"].view(np.uint8), 4096, 6) == 0).all()
    tsv = str(tmp_path / "t.tsv")
    r = subprocess.run([hg.CLI_PATH, "dist", "-r", out, "-q", out, "-o", tsv], capture_output=True, text=True)
    assert r.returncode == 0 and open(tsv).read() == ""  # one record, symmetric -> zero pairs
    # more shards than sketches (shards without rows), and the one record against a copy of the file: ANI of two zero vectors
    out2 = str(tmp_path / "t2.sketch")
    hg.write_sketch_file(out2, rec)
    for extra in ([], ["--shards", "3"]):
        r = subprocess.run([hg.CLI_PATH, "dist", "-r", out, "-q", out2, "-o", tsv, "-a", "0"] + extra, capture_output=True, text=True)
        assert r.returncode == 0, r.stderr
        assert open(tsv).read() == "%s\t%s\t0.000\n" % (rec[0]["file_str"], rec[0]["file_str"])  # 0 / 0 -> NaN -> 0 (src/dist.rs:156-157)


def test_cli_search_topn_and_cpu_mode_reader(tmp_path, orc):
    """`search` (a stub in the reference) = top-n references per query; `-D cpu` (default) reads files the way
    needletail does (FASTQ records, blanks dropped) and ignores `-C false` like src/sketch.rs:89; `-D gpu` honours it."""
    import hypergen_amd as hg
    d = tmp_path / "db"
    d.mkdir()
    L = 200_000
    gen = {"a000.fna": 0, "a005.fna": 5, "a020.fna": 20, "b100.fna": 100, "b107.fna": 107}
    for name, g in gen.items():
        write_fasta(str(d / name), orc.synth_genome(g, L)[1:], name)
    db = str(tmp_path / "db.sketch")
    assert subprocess.run([hg.CLI_PATH, "sketch", "-p", str(d), "-o", db, "-s", "100"], capture_output=True).returncode == 0
    qd = tmp_path / "q"
    qd.mkdir()
    # queries: a FASTQ file with blanks in the sequence line and a gapped FASTA of the same genomes
    s3 = bytes(orc.synth_genome(3, L)[1:]).decode()
    (qd / "q003.fna").write_text("@read one\n" + s3[:1000] + " " + s3[1000:] + "\r\n+\n" + "I" * (L + 1) + "\n")
    s104 = bytes(orc.synth_genome(104, L)[1:]).decode()
    (qd / "q104.fa").write_text(">x\n" + "\n".join(s104[i:i + 61] + ("\t" if i % 3 == 0 else "") for i in range(0, L, 61)) + "\n")
    qs = str(tmp_path / "q.sketch")
    r = subprocess.run([hg.CLI_PATH, "sketch", "-p", str(qd), "-o", qs, "-s", "100", "-C", "false"], capture_output=True, text=True)
    assert r.returncode == 0, r.stderr
    recs = hg.read_sketch_file(qs)
    assert [os.path.basename(x["file_str"]) for x in recs] == ["q003.fna", "q104.fa"]
    for x, g in zip(recs, (3, 104)):  # CPU mode: canonical k-mers although -C false was given; flag stored as given
        w_hv, w_n2, _ = orc.sketch_genome(orc.synth_genome(g, L), scaled=100, canonical=True, norm=orc.NORM_U2T)
        assert x["canonical"] is False and x["hv_norm_2"] == w_n2
        assert (hg.hv_unpack(x["hv"].view(np.uint8), 4096, x["hv_quant_bits"]) == w_hv).all()
    out = str(tmp_path / "top.tsv")
    r = subprocess.run([hg.CLI_PATH, "search", "-r", db, "-q", qs, "-o", out, "-a", "80", "-n", "2"], capture_output=True, text=True)
    assert r.returncode == 0, r.stderr
    rows = [l.split("\t") for l in open(out).read().splitlines()]
    names = [(os.path.basename(a), os.path.basename(b)) for a, b, _ in rows]
    # member m carries 0.1 % * m substitutions against its cluster root (members 0 and 100 ARE the roots): genome 3
    # is 0.3 % from a000 and ~0.8 % from a005, genome 104 is 0.4 % from b100 and ~1.1 % from b107
    assert names == [("q003.fna", "a000.fna"), ("q003.fna", "a005.fna"), ("q104.fa", "b100.fna"), ("q104.fa", "b107.fna")]
    anis = [float(v) for _, _, v in rows]
    assert anis[0] >= anis[1] > 98.0 and anis[2] >= anis[3] > 98.0
    hvs = np.stack([hg.hv_unpack(x["hv"].view(np.uint8), 4096, x["hv_quant_bits"]) for x in hg.read_sketch_file(db)])
    n2s = np.array([x["hv_norm_2"] for x in hg.read_sketch_file(db)], np.int32)
    qh = np.stack([hg.hv_unpack(x["hv"].view(np.uint8), 4096, x["hv_quant_bits"]) for x in recs])
    qn = np.array([x["hv_norm_2"] for x in recs], np.int32)
    model = orc.ani_matrix(hvs, n2s, qh, qn, 21)
    assert rows[0][2] == "%.3f" % float(model[0, 0]) and rows[2][2] == "%.3f" % float(model[3, 1])  # the oracle's text
    # -D gpu honours -C false (src/cuda_kernel.cu:312-314) and reads lines as they are (plain FASTA only)
    gq = tmp_path / "gq"
    gq.mkdir()
    write_fasta(str(gq / "g.fna"), orc.synth_genome(3, L)[1:], "g")
    gs = str(tmp_path / "g.sketch")
    r = subprocess.run([hg.CLI_PATH, "sketch", "-p", str(gq), "-o", gs, "-s", "100", "-C", "false", "-D", "gpu"], capture_output=True, text=True)
    assert r.returncode == 0, r.stderr
    x = hg.read_sketch_file(gs)[0]
    w_hv, w_n2, _ = orc.sketch_genome(orc.synth_genome(3, L), scaled=100, canonical=False)
    assert x["hv_norm_2"] == w_n2 and (hg.hv_unpack(x["hv"].view(np.uint8), 4096, x["hv_quant_bits"]) == w_hv).all()
    # without -r / -q / -o `search` stays the reference's no-op
    assert subprocess.run([hg.CLI_PATH, "search"], capture_output=True).returncode == 0


def test_cli_hv_d_not_a_multiple_of_256_follows_the_reference(tmp_path, orc):
    """`-d 1000`: the reference packs hv_d / 256 whole blocks into q * (hv_d >> 3) bytes and never says that the last
    232 dimensions are lost (src/hd.rs:143-153); `dist` then sees -2^(q-1) there (src/hd.rs:194,206-212) while the
    stored norm is that of the full vector.  Same bytes, same ANI here (the oracle restates both)."""
    import hypergen_amd as hg
    d = tmp_path / "fa"
    d.mkdir()
    gen = {"a.fna": 0, "b.fna": 10, "c.fna": 101}
    for name, g in gen.items():
        write_fasta(str(d / name), orc.synth_genome(g, 200_000)[1:], name)
    files = sorted(str(d / n) for n in gen)
    out = str(tmp_path / "d1000.sketch")
    r = subprocess.run([hg.CLI_PATH, "sketch", "-p", str(d), "-o", out, "-s", "100", "-d", "1000"], capture_output=True, text=True)
    assert r.returncode == 0 and "not a multiple of 256" in (r.stdout + r.stderr), r.stderr
    recs = hg.read_sketch_file(out)
    hvs, n2s = [], []
    for x, f in zip(recs, files):
        hv, n2, _ = orc.sketch_genome(hg.read_merge_seq(f), scaled=100, hv_d=1000, norm=orc.NORM_U2T)
        q, packed = orc.pack_hv(hv)
        assert x["hv_d"] == 1000 and x["hv_quant_bits"] == q and x["hv_norm_2"] == n2
        assert packed.size == q * 125 and x["hv"].size == packed.size // 2
        assert (x["hv"].view(np.uint8) == packed[: 2 * x["hv"].size]).all()
        hvs.append(orc.unpack_hv(packed, 1000, q))  # what the reference's dist works on
        n2s.append(n2)
        assert (hvs[-1][768:] == -(1 << (q - 1))).all() and (hvs[-1][:768] == hv[:768]).all()
    tsv = str(tmp_path / "ani.tsv")
    r = subprocess.run([hg.CLI_PATH, "dist", "-r", out, "-q", out, "-o", tsv, "-a", "0"], capture_output=True, text=True)
    assert r.returncode == 0, r.stderr
    want = orc.ani_matrix(np.stack(hvs), np.array(n2s, np.int32), np.stack(hvs), np.array(n2s, np.int32), 21)
    got = {(a, b): v for a, b, v in (l.split("\t") for l in open(tsv).read().splitlines())}
    assert len(got) == 3
    for i in range(3):
        for j in range(i + 1, 3):
            assert got[(files[i], files[j])] == "%.3f" % float(want[i, j])


def test_device_unpack_matches_the_host_decoders(orc):
    """hg_hv_unpack_batch_dev (decompress_file_sketch on the device, src/hd.rs:171-232): both payload layouts, every width,
    payloads at odd byte offsets (they sit behind path strings in a file image), hv_d with a partial last block -- the same
    integers as the oracle's decoders, the reference's lossy corners (q = 16, -2^(q-1) in the naive layout) included."""
    import torch
    import hypergen_amd as hg
    rng = np.random.default_rng(41)
    dev = torch.device("cuda:0")
    with hg.Context(0) as ctx:
        for hv_d in (4096, 1000, 256, 16384):
            blobs, offs, qs, lays, want = [], [], [], [], []
            pos = 0
            for i in range(64):
                q = 6 + i % 11
                lim = 1 << (q - 1)
                hv = rng.integers(-lim, lim, hv_d).astype(np.int16)
                hv[:2] = [-lim, lim - 1]
                naive = i % 3 == 1
                if naive:
                    _, w = orc.pack_hv_naive(hv, q)
                    payload = w.view(np.uint8)
                    want.append(orc.unpack_hv_naive(w, hv_d, q))
                else:
                    _, payload = orc.pack_hv(hv, q)
                    payload = payload[: payload.size // 2 * 2]  # the i16 view the file stores (src/hd.rs:155-157)
                    want.append(orc.unpack_hv(np.concatenate([payload, np.zeros(1, np.uint8)]), hv_d, q))
                pad = rng.integers(0, 5)  # any alignment
                blobs += [np.full(pad, 0xEE, np.uint8), payload]
                pos += pad
                offs.append(pos), qs.append(q), lays.append(hg.PAYLOAD_NAIVE if naive else hg.PAYLOAD_BITPACKER8X)
                pos += payload.size
            img = torch.from_numpy(np.concatenate(blobs)).to(dev)
            out = torch.full((64, hv_d), 12345, dtype=torch.int16, device=dev)
            ctx.hv_unpack_batch_dev(img.data_ptr(), img.numel(), offs, qs, lays, hv_d, out.data_ptr())
            got = out.cpu().numpy()
            for i in range(64):
                assert np.array_equal(got[i], want[i]), (hv_d, i, qs[i], lays[i])
            # a payload that reaches past the buffer is refused, not read
            with pytest.raises(hg.HgError):
                ctx.hv_unpack_batch_dev(img.data_ptr(), offs[-1] + 8, offs, qs, lays, hv_d, out.data_ptr())
            # a buffer that itself starts at an odd address with the first payload at offset 0 and the last one ending with
            # the buffer: the kernel's aligned staging loads may touch neither the bytes in front nor the bytes behind
            for shift in (1, 2, 3):
                body = np.concatenate(blobs)[offs[0]:]                      # from the first payload on
                buf = torch.zeros(shift + body.size, dtype=torch.uint8, device=dev)
                buf[shift:] = torch.from_numpy(body).to(dev)
                rel = [o - offs[0] for o in offs]
                out.fill_(777)
                ctx.hv_unpack_batch_dev(buf.data_ptr() + shift, body.size, rel, qs, lays, hv_d, out.data_ptr())
                got = out.cpu().numpy()
                for i in (0, 1, 62, 63):
                    assert np.array_equal(got[i], want[i]), (hv_d, shift, i)


def test_cli_reads_and_writes_the_non_avx2_payload_layout(tmp_path, orc):
    """`hyper-gen sketch --pack_layout naive` writes what a reference host without AVX2 writes (src/hd.rs:158-166); dist and
    search read either layout (told apart by the payload length), also mixed in one file.  The naive decode is the
    reference's: a sketch holding the value -2^(q-1) comes back with +2^(q-1) there."""
    import hypergen_amd as hg
    d = tmp_path / "fa"
    d.mkdir()
    names = ["a.fna", "b.fna", "c.fna", "d.fna"]
    for i, n in enumerate(names):
        write_fasta(str(d / n), orc.synth_genome(i * 7, 200_000)[1:], n)
    out_a, out_n = str(tmp_path / "avx2.sketch"), str(tmp_path / "naive.sketch")
    for out, extra in ((out_a, []), (out_n, ["--pack_layout", "naive"])):
        r = subprocess.run([hg.CLI_PATH, "sketch", "-p", str(d), "-o", out, "-s", "100"] + extra, capture_output=True, text=True)
        assert r.returncode == 0, r.stderr
    ra, rn = hg.read_sketch_file(out_a), hg.read_sketch_file(out_n)
    hvs = []
    for x, y in zip(ra, rn):
        q = x["hv_quant_bits"]
        hv = hg.hv_unpack(x["hv"].view(np.uint8), 4096, q)
        hvs.append(hv)
        assert y["hv_quant_bits"] == q and y["hv_norm_2"] == x["hv_norm_2"] and y["hv"].size == (q * 4096 + 16) // 16
        assert np.array_equal(y["hv"], orc.pack_hv_naive(hv, q)[1])
    # dist on the naive file == the oracle's ANI of what the reference's naive decoder returns for these payloads
    dec = np.stack([orc.unpack_hv_naive(y["hv"], 4096, y["hv_quant_bits"]) for y in rn])
    n2 = np.array([y["hv_norm_2"] for y in rn], np.int32)
    want = orc.ani_matrix(dec, n2, dec, n2, 21)
    mixed = str(tmp_path / "mixed.sketch")
    hg.write_sketch_file(mixed, [ra[0], rn[1], ra[2], rn[3]])
    for sk in (out_n, mixed):
        tsv = str(tmp_path / "o.tsv")
        r = subprocess.run([hg.CLI_PATH, "dist", "-r", sk, "-q", out_a, "-o", tsv, "-a", "0"], capture_output=True, text=True)
        assert r.returncode == 0, r.stderr
        rows = [l.split("\t") for l in open(tsv).read().splitlines()]
        assert len(rows) == 16
        idx = {str(d / n): i for i, n in enumerate(names)}
        for a, b, v in rows:
            i, j = idx[a], idx[b]
            ref_row = dec[i] if (sk == out_n or i in (1, 3)) else hvs[i]
            w = orc.ani_matrix(ref_row[None], n2[i: i + 1], hvs[j][None], n2[j: j + 1], 21)[0, 0]
            assert v == "%.3f" % float(w), (sk, a, b)
    assert want.shape == (4, 4)
    # a payload of neither length is refused
    bad = dict(ra[0])
    bad["hv"] = ra[0]["hv"][:-3]
    hg.write_sketch_file(mixed, [bad])
    r = subprocess.run([hg.CLI_PATH, "dist", "-r", mixed, "-q", out_a, "-o", str(tmp_path / "x.tsv")], capture_output=True, text=True)
    assert r.returncode != 0 and "corrupt sketch payload" in r.stderr


def oracle_tsv(orc, hv_r, n2r, hv_q, n2q, names_r, names_q, k, ani_th, sym, ani=None):
    """What the reference's dist writes (src/dist.rs:231-294 + src/utils.rs:260-308) from the ORACLE's ANI values: pairs in
    enumeration order (row-major, i < j only when the two files are the same), stable ascending sort by ANI, reversed, printed
    while ani >= ani_th as "{}\\t{}\\t{:.3}"."""
    full = orc.ani_matrix(hv_r, n2r, hv_q, n2q, k) if ani is None else ani
    R, Q = full.shape
    if sym:
        ii, jj = np.triu_indices(R, 1)  # (row-major, like the reference's nested loops)
    else:
        ii, jj = np.divmod(np.arange(R * Q), Q)
    v = full[ii, jj]
    keep = v >= np.float32(ani_th)
    ii, jj, v = ii[keep], jj[keep], v[keep]
    order = np.argsort(v, kind="stable")[::-1]
    return "".join("%s\t%s\t%.3f\n" % (names_r[ii[t]], names_q[jj[t]], float(v[t])) for t in order)


def test_cli_dist_tsv_is_the_library_hits_in_dump_order_with_exact_formatting(tmp_path, orc):
    """`hyper-gen dist` on two different files: the TSV must be, byte for byte, the ORACLE's lines (oracle_tsv: its ANI values,
    the reference's enumeration, order and format) -- and the library's own hits (hg_dist) in
    dump_ani_file's order (hg_sort_ani_hits) printed as "{}\\t{}\\t{:.3}" -- the CLI formats without printf (ties at the third
    decimal round to even, as Rust's and glibc's exact formatting do), decodes the payloads on the device and, with one
    device, orders the hits there before they leave it"""
    import torch
    import bench
    import hypergen_amd as hg
    dev = torch.device("cuda:0")
    n = 700
    a = bench.clustered_hvs(n, 0, dev, n=900).cpu().numpy()
    b = bench.clustered_hvs(n, 0, dev, n=900, salt=1).cpu().numpy()
    a[5] = a[4]  # identical rows: ANI exactly 100 and ties in the order
    paths = []
    for name, hv in (("a", a), ("b", b)):
        recs = []
        for i in range(n):
            q, pk = hg.hv_pack(hv[i])
            recs.append(dict(ksize=21, scaled=1500, canonical=True, seed=123, hv_d=4096, hv_quant_bits=q,
                             hv_norm_2=int((hv[i].astype(np.int64) ** 2).sum()), file_str="/d/%s/%s%04d.fna" % (name, "x" * (i % 3), i),
                             hv=pk.view(np.int16)))
        p = str(tmp_path / (name + ".sketch"))
        hg.write_sketch_file(p, recs)
        paths.append((p, [r["file_str"] for r in recs], np.array([r["hv_norm_2"] for r in recs], np.int32)))
    with hg.Context(0) as ctx:
        for (pr, nr, n2r), (pq, nq, n2q), hv_r, hv_q, sym in ((paths[0], paths[1], a, b, False), (paths[0], paths[0], a, a, True)):
            hits = ctx.dist(hv_r, n2r, hv_q, n2q, 21, symmetric=sym, ani_th=80.0)
            hits = hg.sort_ani_hits(hits, n, symmetric=sym)
            want = "".join("%s\t%s\t%.3f\n" % (nr[h["ref_idx"]], nq[h["qry_idx"]], float(h["ani"])) for h in hits)
            assert want == oracle_tsv(orc, hv_r, n2r, hv_q, n2q, nr, nq, 21, 80.0, sym)  # ... which are the ORACLE's lines
            # one shard per visible GPU (the hits never touch the host unordered), and three shards dealt round the GPUs: the
            # several-GPU path (row blocks decoded per shard, peer pulls, per-shard hit lists merged and ordered through
            # device 0) on however many GPUs the box has
            for extra in ([], ["--shards", "3"]):
                tsv = str(tmp_path / "o.tsv")
                r = subprocess.run([hg.CLI_PATH, "dist", "-r", pr, "-q", pq, "-o", tsv, "-a", "80"] + extra, capture_output=True, text=True)
                assert r.returncode == 0, r.stderr
                got = open(tsv).read()
                assert len(hits) > 3000 and got == want, (sym, extra, len(hits))
        # search: top 3 references per query, the same through one shard and through two
        outs = []
        for extra in ([], ["--shards", "2"]):
            tsv = str(tmp_path / "s.tsv")
            r = subprocess.run([hg.CLI_PATH, "search", "-r", paths[0][0], "-q", paths[1][0], "-o", tsv, "-a", "80", "-n", "3"] + extra,
                               capture_output=True, text=True)
            assert r.returncode == 0, r.stderr
            outs.append(open(tsv).read())
        assert outs[0] == outs[1] and outs[0].count("\n") >= n
