"""The CPU oracle against the golden vectors -- no GPU."""
import numpy as np
import pytest

from conftest import golden


def test_t1ha2_upstream_selfcheck(orc):
    g = golden("kat_t1ha2.json")
    pat = bytes(g["pattern"])
    for c in g["cases"]:
        assert orc.t1ha2_atonce(pat[: c["len"]], int(c["seed"])) == int(c["hash"], 16), c


def test_t1ha2_whole_upstream_table_and_dna_beyond_32_bytes(orc):
    """all 81 entries of upstream's t1ha_refval_2atonce (47 inputs longer than 32 bytes: the loop src/cuda_kernel.cu:196-246
    omits and src/sketch.rs:90 reaches through the t1ha crate for k > 32), and DNA strings of every length 33..255 hashed by
    tools/gen_golden_cpu.py's own Python-integer t1ha2 (which reproduces the 81 first)"""
    g = golden("kat_t1ha2.json")
    assert len(g["upstream_selfcheck"]) == 81 and sum(len(c["data"]) // 2 > 32 for c in g["upstream_selfcheck"]) == 47
    for c in g["upstream_selfcheck"]:
        assert orc.t1ha2_atonce(bytes.fromhex(c["data"]), int(c["seed"])) == int(c["hash"], 16), c["name"]
    assert [len(c["seq"]) for c in g["dna_33_255"]] == list(range(33, 256))
    for c in g["dna_33_255"]:
        sq = c["seq"].encode()
        rc = sq.translate(bytes.maketrans(b"ACGT", b"TGCA"))[::-1]
        assert orc.t1ha2_atonce(sq, int(c["seed"])) == int(c["hash"], 16), len(sq)
        assert orc.t1ha2_atonce(rc, int(c["seed"])) == int(c["hash_revcomp"], 16), len(sq)
        # ... and through the k-mer walk: one k-mer of length k, both strand modes (src/sketch.rs:89: the smaller string)
        arr = np.frombuffer(sq, np.uint8)
        assert [int(x) for x in orc.kmer_hash_sample(arr, len(sq), 1, 123, False)] == [int(c["hash"], 16)]
        want = int(c["hash"], 16) if sq <= rc else int(c["hash_revcomp"], 16)
        assert [int(x) for x in orc.kmer_hash_sample(arr, len(sq), 1, 123, True)] == [want]


def test_wyrng_upstream_kat(orc):
    g = golden("kat_wyrng.json")
    assert orc.wyrng_stream(g["seed"], 1)[0] == int(g["first"], 16) == 0x3E99A772750DCBE  # wyhash crate README
    # 16 consecutive outputs for 8 seeds from tools/gen_golden_wyrng.py (the published definition in Python integers,
    # written independently of hg_oracle.c)
    assert len(g["streams"]) >= 4
    for st in g["streams"]:
        want = [int(x, 16) for x in st["next_u64"]]
        assert len(want) >= 16 and orc.wyrng_stream(int(st["seed"], 16), len(want)) == want, st["seed"]
    # seed_from_u64 is the identity on the state: successive states differ by the increment
    a = orc.wyrng_stream(5, 3)
    assert a[1:] == orc.wyrng_stream((5 + 0xA0761D6478BD642F) % 2**64, 2)


def test_g1_reference_fixture(orc):
    g = golden("g1_test_fna.json")
    seq = orc.read_merge_seq(g["fasta"].encode())
    assert bytes(seq) == b"NAGCTCTTANNAGCCCNTTacgttacagccctgaaaacttt"
    for key, k, scaled in (("k21_scaled1", 21, 1), ("k5_scaled1", 5, 1), ("k21_scaled1500", 21, 1500)):
        got = orc.kmer_hash_sample(seq, k, scaled, g["seed"], g["canonical"])
        assert ["%016x" % int(x) for x in got] == g[key], key
    # config 1 end to end: empty set -> zero HV, norm 0, 6-bit payload of the constant 32
    hv, n2, nh = orc.sketch_genome(seq)
    assert nh == 0 and n2 == 0 and not hv.any()
    q, packed = orc.pack_hv(hv)
    assert q == 6 and packed.size == 6 * 4096 // 8
    assert (orc.unpack_hv(packed, 4096, 6) == 0).all()
    w = np.frombuffer(packed.tobytes(), "<u4")[:6]  # 32 x (32 at 6 bits) per lane
    v = sum(32 << (6 * r) for r in range(32))
    assert [int(x) for x in w[:1]] == [v & 0xFFFFFFFF]


def test_read_merge_seq_layout(orc):
    txt = b">a desc\nACGT\r\nAC\n>b\nGG\n\nTT"
    assert bytes(orc.read_merge_seq(txt)) == b"NACGTACNGGTT"
    assert orc.read_merge_seq(b"").size == 0


def test_kmer_edge_cases(orc):
    assert orc.kmer_hash_sample(b"", 21, 1).size == 0
    assert orc.kmer_hash_sample(b"ACGT" * 5, 21, 1).size == 0  # 20 bases < k
    one = orc.kmer_hash_sample(b"ACGTACGTACGTACGTACGTA", 21, 1)
    assert one.size == 1
    # strand symmetry: a sequence and its reverse complement sample the same set
    rng = np.random.default_rng(1)
    s = rng.choice(np.frombuffer(b"ACGT", np.uint8), 5000)
    comp = np.zeros(256, np.uint8)
    comp[list(b"ACGT")] = list(b"TGCA")
    rc = comp[s[::-1]]
    a, b = orc.kmer_hash_sample(s, 21, 50), orc.kmer_hash_sample(rc, 21, 50)
    assert a.size > 50 and (a == b).all()
    # non-canonical mode differs, lower case is folded, N breaks runs, U only in U2T mode
    assert not np.array_equal(orc.kmer_hash_sample(s, 21, 50, canonical=False), a)
    assert (orc.kmer_hash_sample(np.char.lower(s.view("S1")).view(np.uint8), 21, 50) == a).all()
    t = s.copy()
    t[100] = ord("N")
    assert orc.kmer_hash_sample(t, 21, 1, unique=False).size == 5000 - 20 - 21
    u = s.copy()
    u[s == ord("T")] = ord("U")
    assert (orc.kmer_hash_sample(u, 21, 50, norm=orc.NORM_U2T) == a).all()
    assert orc.kmer_hash_sample(u, 21, 50, norm=orc.NORM_ACGT).size < a.size
    # duplicates are reported by the walk and removed by the set
    rep = np.tile(s[:300], 4)
    d = orc.kmer_hash_sample(rep, 21, 10, unique=False)
    assert np.unique(d).size < d.size


def test_k_over_32(orc):
    rng = np.random.default_rng(2)
    s = rng.choice(np.frombuffer(b"ACGT", np.uint8), 400)
    hs = orc.kmer_hash_sample(s, 41, 1, unique=False)
    assert hs.size == 360
    assert int(hs[0]) in (orc.t1ha2_atonce(s[:41], 123),
                          orc.t1ha2_atonce(bytes(s[:41]).translate(bytes.maketrans(b"ACGT", b"TGCA"))[::-1], 123))


@pytest.mark.parametrize("n", [0, 1, 3, 4, 7, 50])
@pytest.mark.parametrize("d", [256, 4096])
def test_hv_layouts(orc, n, d):
    rng = np.random.default_rng(n * 1000 + d)
    hs = np.unique(rng.integers(0, 2**63, n, dtype=np.uint64))
    sc = orc.encode_hv(hs, d, orc.LAYOUT_SCALAR)
    av = orc.encode_hv(hs, d, orc.LAYOUT_AVX2)
    # closed-form permutation == emulated intrinsic sequence of src/hd.rs:14-92
    assert (av == orc.encode_hv_avx2_emulated(hs, d)).all()
    j = np.arange(64)
    perm = (np.arange(d) // 64 * 64)[:, None].reshape(-1, 64)[:, :1] + (4 * (j & 15) + (j >> 4))[None, :]
    assert (av[perm.reshape(-1)] == sc).all()
    # definition: hv = 2*count - n
    cnt = np.zeros(d, np.int64)
    for h in hs:
        words = orc.wyrng_stream(int(h), d // 64)
        bits = np.unpackbits(np.array(words, "<u8").view(np.uint8), bitorder="little")
        cnt += bits
    assert (sc == 2 * cnt - hs.size).all()
    # src/lib.rs:224 intent: dot products are layout invariant
    assert orc.hv_norm2(sc) == orc.hv_norm2(av) == int((sc.astype(np.int64) ** 2).sum())


def test_avx2_layout_with_the_real_intrinsics(orc):
    """src/hd.rs:14-92 followed statement by statement with <immintrin.h> (oracle/hg_oracle_avx2.c, gcc -mavx2):
    the CPU's own vpshufb / vphaddw / vpermq define the AVX2 output order; the closed-form ORC_LAYOUT_AVX2 and
    the scalar emulation must both equal it -- including the zero-padded last batch (n % 4 != 0), n = 0, a
    production-sized set and the i16 wrap of n > 32 767."""
    if not orc.has_avx2_build():
        pytest.skip("host CPU without AVX2 (or libhg_oracle_avx2.so not built)")
    rng = np.random.default_rng(11)
    for n in (0, 1, 2, 3, 4, 5, 7, 8, 50, 3333, 40_000):
        for d in ((256, 4096) if n < 1000 else (4096 if n == 3333 else 256,)):
            hs = np.unique(rng.integers(0, 2**64, n, dtype=np.uint64))
            real = orc.encode_hv_avx2_intrinsics(hs, d)
            assert (real == orc.encode_hv(hs, d, orc.LAYOUT_AVX2)).all(), (n, d)
            if n <= 3333:
                assert (real == orc.encode_hv_avx2_emulated(hs, d)).all(), (n, d)
    # the permutation itself, on a single hash: scalar dim 64i + j  ->  AVX2 dim 64i + 4*(j & 15) + (j >> 4)
    h = np.array([0x0123456789ABCDEF], np.uint64)
    sc, av = orc.encode_hv(h, 256, orc.LAYOUT_SCALAR), orc.encode_hv_avx2_intrinsics(h, 256)
    for i in range(4):
        for j in range(64):
            assert av[64 * i + 4 * (j & 15) + (j >> 4)] == sc[64 * i + j]


def test_pack_roundtrip_and_width(orc):
    rng = np.random.default_rng(3)
    for amp in (5, 31, 32, 33, 200, 600, 5000, 16000):
        hv = rng.integers(-amp, amp + 1, 4096).astype(np.int16)
        q, packed = orc.pack_hv(hv)
        lo, hi = int(hv.min()), int(hv.max())
        assert -(1 << (q - 1)) <= lo and hi <= (1 << (q - 1)) - 1
        assert q == 6 or not (-(1 << (q - 2)) <= lo and hi <= (1 << (q - 2)) - 1)
        assert packed.size == q * 4096 // 8
        assert (orc.unpack_hv(packed, 4096, q) == hv).all()  # src/lib.rs:261 intent
    # layout spot check: value r of lane l sits at bit r*q of that lane's stream
    hv = np.zeros(256, np.int16)
    hv[8 * 5 + 3] = 7  # row 5, lane 3
    q, packed = orc.pack_hv(hv, 9)
    words = np.frombuffer(packed.tobytes(), "<u4").reshape(9, 8)
    stream = sum(int(words[w, 3]) << (32 * w) for w in range(9))
    assert (stream >> (5 * 9)) & 0x1FF == 7 + 256


def _bp8x_model(values, q):
    """BitPacker8x block from the crate's definition, bit by bit in Python integers (a third formulation, shared with
    neither oracle/hg_oracle.c nor hg_formats.cpp): lane l = i % 8 owns a stream of 32*q bits, element r = i // 8
    starts at bit r*q; an unmasked element spills up to bit 31 of its first word and, if it straddles a word
    boundary, its bits from (32 - c) upwards continue at bit 0 of the next word; word w of lane l = output u32 8w + l."""
    words = [[0] * 8 for _ in range(q)]
    for i, v in enumerate(values):
        lane, p = i % 8, (i // 8) * q
        w0, c = divmod(p, 32)
        words[w0][lane] |= (v << c) & 0xFFFFFFFF
        if c + q > 32:
            words[w0 + 1][lane] |= v >> (32 - c)
    return np.array(words, dtype="<u4").tobytes()


def test_pack_follows_the_crate_definition_incl_q16_and_partial_blocks(orc):
    rng = np.random.default_rng(11)
    for q in range(6, 17):
        for d in (256, 1024, 100, 264, 1000, 4096 + 8):
            amp = (1 << (q - 1)) - 1
            hv = rng.integers(-amp - 1, amp + 1, d).astype(np.int16)
            _, packed = orc.pack_hv(hv, q)
            assert packed.size == q * (d >> 3)  # src/hd.rs:146
            off = -32768 if q == 16 else 1 << (q - 1)  # src/hd.rs:140: i16
            want = b""
            for b in range(d // 256):
                # (i + offset) as u32: wrapping i16 sum, sign-extended
                blk = [(((int(x) + off + 32768) % 65536) - 32768) & 0xFFFFFFFF for x in hv[256 * b: 256 * b + 256]]
                want += _bp8x_model(blk, q)
            want += bytes(packed.size - len(want))  # the bytes behind the last whole block stay zero
            assert packed.tobytes() == want, (q, d)
            back = orc.unpack_hv(packed, d, q)
            whole = d // 256 * 256
            if q < 16:
                assert (back[:whole] == hv[:whole]).all()
            elif whole:  # every even-indexed element of a lane that was >= 0 spills 0xFFFF into its odd neighbour
                assert (back[:whole] != hv[:whole]).any()
            assert (back[whole:] == np.int16(-off if q < 16 else -32768)).all()  # 0 as i16 - offset (src/hd.rs:206-212)


def _naive_model(hv, q):
    """third statement of the non-AVX2 layout (src/hd.rs:158-166), on Python integers: the stream is the concatenation of
    the values' low q bits, LSB first; 16 stream bits per i16 word, (q*d + 16) // 16 words"""
    bits = []
    for x in hv:
        bits += [((int(x) & 0xFFFF) >> k) & 1 for k in range(q)]
    words = [0] * ((q * len(hv) + 16) // 16)
    for i, b in enumerate(bits):
        words[i // 16] |= b << (i % 16)
    return np.array(words, np.uint16).view(np.int16)


def test_naive_layout_follows_the_reference_loops(orc):
    """the payload of hosts without AVX2 (src/hd.rs:158-166, 213-231): oracle against the bit-level model, and the
    reference's decode quirks -- strict `>` (the value -2^(q-1) comes back as +2^(q-1)) and the i16 shifts at q = 16"""
    rng = np.random.default_rng(12)
    for q in range(6, 17):
        for d in (16, 256, 100, 1000, 4096):
            lim = 1 << (q - 1)
            hv = rng.integers(-lim, lim, d).astype(np.int16)
            hv[:3] = [-lim, lim - 1, -1]
            _, words = orc.pack_hv_naive(hv, q)
            assert words.size == (q * d + 16) // 16 and np.array_equal(words, _naive_model(hv, q)), (q, d)
            back = orc.unpack_hv_naive(words, d, q)
            if q < 16:
                want = hv.astype(np.int32)
                want[want == -lim] = lim  # low bits 100..0: `v > 1 << (q-1)` is false, the value stays +2^(q-1)
                assert np.array_equal(back, want.astype(np.int16)), (q, d)
            else:  # `1 << 15` = -32768, `1 << 16` wraps to 1: v > -32768 -> v - 1
                want = np.where(hv == -32768, hv, (hv.astype(np.int32) - 1).astype(np.int16))
                assert np.array_equal(back, want), d


def test_ani_golden(orc):
    for c in golden("g4_ani.json"):
        got = orc.ani_from_dot(c["dot"], c["nr"], c["nq"], c["k"])
        assert np.float32(got) == np.float32(c["ani"]), c  # equal: the fixture's logarithm is glibc's algorithm in Python doubles
    assert orc.ani_from_dot(0, 10, 10) == 0.0 and orc.ani_from_dot(-3, 10, 10) == 0.0
    assert orc.ani_from_dot(10, 10, 10) == 100.0


def test_host_logf_is_the_restated_glibc_algorithm(orc):
    """The oracle's ANI calls the host's logf (what Rust's f32::ln calls).  It must be glibc's table-driven routine, in either
    of its two builds (fused / unfused multiply-adds: they agree on all 2^32 inputs) -- the device restates that algorithm
    (hyper-gen_amd/csrc/hg_logf.h), so a host with another libm would make "equal to the oracle" mean something else.
    2^27 bit patterns around 1.0, where 2 / (1 / J + 1) lives, + subnormals + a stride over the rest."""
    for fused in (1, 0):
        assert orc.logf_sweep(0x3F000000 - (1 << 26), 1 << 27, fused)[0] == 0
        assert orc.logf_sweep(0, 1 << 16, fused)[0] == 0 and orc.logf_sweep(0x7F7F0000, 1 << 17, fused)[0] == 0
    x = np.arange(0, 1 << 32, 4099, dtype=np.uint64).astype(np.uint32).view(np.float32)
    h, f, u = orc.logf_array(x, -1), orc.logf_array(x, 1), orc.logf_array(x, 0)
    nan = np.isnan(h)
    assert (np.isnan(f) == nan).all() and (np.isnan(u) == nan).all()
    assert (h.view(np.uint32)[~nan] == f.view(np.uint32)[~nan]).all() and (h.view(np.uint32)[~nan] == u.view(np.uint32)[~nan]).all()


def test_ani_matrix_matches_pairwise(orc):
    rng = np.random.default_rng(4)
    r = rng.integers(-200, 200, (5, 512)).astype(np.int16)
    q = np.vstack([r[:2], rng.integers(-200, 200, (2, 512)).astype(np.int16)])
    rn = np.array([orc.hv_norm2(x) for x in r], np.int32)
    qn = np.array([orc.hv_norm2(x) for x in q], np.int32)
    m = orc.ani_matrix(r, rn, q, qn, 21)
    assert m.shape == (5, 4) and m[0, 0] == 100.0 and m[1, 1] == 100.0
    dot = int(r[3].astype(np.int64) @ q[2].astype(np.int64))
    assert m[3, 2] == orc.ani_from_dot(dot, rn[3], qn[2])


def test_synth_genomes(orc):
    root = orc.synth_genome(0, 20000)
    assert root[0] == ord("N") and set(root[1:].tolist()) <= set(b"ACGT")
    assert (orc.synth_genome(0, 5000)[:5001] == root[:5001]).all()  # counter based
    m50 = orc.synth_genome(50, 20000)
    rate = float((m50[1:] != root[1:]).mean())
    assert 0.04 < rate < 0.06
    other = orc.synth_genome(100, 20000)
    assert 0.70 < float((other[1:] != root[1:]).mean()) < 0.80
    counts = np.bincount(root[1:], minlength=128)[list(b"ACGT")] / 20000
    assert (abs(counts - 0.25) < 0.02).all()


def test_g2_reference_kernel_vectors(orc):
    """Hash sets produced ON THE MI355X by the reference's own kernel (src/cuda_kernel.cu compiled in
    place by hipcc, tools/gen_golden_ref_gpu.py) -- the oracle must reproduce them on the CPU."""
    cases = golden("g2_ref_kernel.json")
    assert len(cases) >= 8
    for c in cases:
        if c["L"] > 1_000_000:
            continue  # the 5 Mbp case runs in test_g2_reference_kernel_5mbp
        _check_g2(orc, c)


def test_g2_reference_kernel_5mbp(orc):
    big = [c for c in golden("g2_ref_kernel.json") if c["L"] > 1_000_000]
    assert big
    for c in big:
        _check_g2(orc, c)


def _check_g2(orc, c):
    seq = orc.synth_genome(c["genome"], c["L"])
    if c["mutated"]:
        rng = np.random.default_rng(c["genome"])
        seq = seq.copy()
        seq[rng.choice(c["L"], 50, replace=False) + 1] = ord("N")
        seq[5000:6000] = np.char.lower(seq[5000:6000].view("S1")).view(np.uint8)
    got = orc.kmer_hash_sample(seq, c["k"], c["scaled"], 123, c["canonical"])
    assert got.size == c["n"], c["name"]
    assert "%016x" % int(np.bitwise_xor.reduce(got) if got.size else 0) == c["xor"], c["name"]
    assert "%016x" % (int(got.astype(object).sum()) % 2**64 if got.size else 0) == c["sum"], c["name"]
    want = np.array([int(h, 16) for h in c["hashes"]], np.uint64)
    if c["subsampled"]:
        assert np.isin(want, got).all(), c["name"]
    else:
        assert (want == got).all(), c["name"]
