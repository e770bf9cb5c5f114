"""Parity of the HIP path (through the C ABI) against the CPU oracle and the golden fixtures.

Bit-exact for hash sets, HV integers and norms -- and for ANI (conftest.ANI_TOL = 0: the device evaluates glibc's logf,
the oracle calls it; BASELINE.json's north_star would allow 1e-4).
"""
import numpy as np
import pytest

from conftest import ANI_TOL, golden

pytestmark = pytest.mark.gpu

ACGT = np.frombuffer(b"ACGT", np.uint8)


@pytest.fixture(scope="module")
def hg():
    import hypergen_amd
    return hypergen_amd


@pytest.fixture(scope="module", params=["ascii", "packed"])
def ctx(hg, request):
    """every test of this module that sketches runs in both input forms: "packed" makes the library 2-bit pack each ASCII
    batch on the device and run the packed-input kernels (hg_ctx_set_debug "kmer_input"); conftest drops the second form
    for the tests that never touch a sequence (dist, Hamming, encode)"""
    c = hg.Context(0)
    c.set_debug("kmer_input", "packed" if request.param == "packed" else "")
    yield c
    c.close()


def rand_seq(rng, n):
    return rng.choice(ACGT, n)


# ---- k-mer hash + sample ---------------------------------------------------------------------
def test_g1_reference_fixture(ctx, orc):
    g = golden("g1_test_fna.json")
    seq = orc.read_merge_seq(g["fasta"].encode())
    for key, k, scaled in (("k21_scaled1", 21, 1), ("k5_scaled1", 5, 1), ("k21_scaled1500", 21, 1500)):
        got = ctx.kmer_hash_sample(seq, k, scaled, g["seed"], g["canonical"])
        assert ["%016x" % int(x) for x in got] == g[key], key


@pytest.mark.parametrize("k", [1, 4, 8, 9, 12, 15, 16, 17, 18, 19, 20, 21, 22, 24, 25, 28, 29, 30, 31, 32])
def test_hash_sets_all_k(ctx, orc, k):
    rng = np.random.default_rng(100 + k)
    s = rand_seq(rng, 20011)
    s[[5, 777, 778, 5000, 12345, 20010]] = ord("N")
    s[3000:3300] = np.char.lower(s[3000:3300].view("S1")).view(np.uint8)
    for canonical in (True, False):
        want = orc.kmer_hash_sample(s, k, 7, 123, canonical)
        got = ctx.kmer_hash_sample(s, k, 7, 123, canonical)
        assert got.size == want.size and (got == want).all(), (k, canonical)


def test_long_k_dna_vectors_of_every_length_33_to_255(ctx):
    """tests/golden/kat_t1ha2.json "dna_33_255": one DNA string per length, hashed by a Python-integer t1ha2 that reproduces
    upstream's 81 self-check values -- through kmer_sample_long as a sequence of exactly one k-mer, both strand modes"""
    g = golden("kat_t1ha2.json")
    tr = bytes.maketrans(b"ACGT", b"TGCA")
    for c in g["dna_33_255"]:
        sq = c["seq"].encode()
        k = len(sq)
        arr = np.frombuffer(sq, np.uint8)
        assert [int(x) for x in ctx.kmer_hash_sample(arr, k, 1, 123, False)] == [int(c["hash"], 16)], k
        want = int(c["hash"], 16) if sq <= sq.translate(tr)[::-1] else int(c["hash_revcomp"], 16)
        assert [int(x) for x in ctx.kmer_hash_sample(arr, k, 1, 123, True)] == [want], k


@pytest.mark.parametrize("k", [33, 40, 41, 51, 63, 64, 65, 96, 97, 128, 255])
def test_hash_sets_long_k(ctx, orc, k):
    """k > 32: t1ha2's long-input loop on the device (the CPU path supports it, src/cuda_kernel.cu does not)."""
    rng = np.random.default_rng(1000 + k)
    s = rand_seq(rng, 9000)
    s[[17, 4000, 4001, 8999]] = ord("N")
    s[2000:2100] = np.char.lower(s[2000:2100].view("S1")).view(np.uint8)
    for canonical in (True, False):
        want = orc.kmer_hash_sample(s, k, 5, 123, canonical)
        got = ctx.kmer_hash_sample(s, k, 5, 123, canonical)
        assert want.size > 100 and got.size == want.size and (got == want).all(), (k, canonical)


@pytest.mark.parametrize("n", [0, 1, 20, 21, 22, 31, 32, 33, 43, 44, 45, 55, 56, 57, 76, 77, 3071, 3072, 3073, 3092, 3093,
                               9215, 9216, 9217, 9236, 9237, 9271, 9272, 24575, 24576, 24577, 24596, 24597,
                               27647, 27648, 27649, 27668, 27669, 27703, 27704, 50000])
def test_hash_sets_ragged_lengths(ctx, orc, n):
    """lengths around the tile / work-item / window edges of both k = 21 kernels (32-base windows: 3 072 and 24 576
    starts; grouped 56-base windows: 9 216 and 27 648 starts), with and without a non-base near the end"""
    rng = np.random.default_rng(n)
    s = rand_seq(rng, n)
    for dirty in (False, True):
        if dirty and n > 30:
            s[n - 1 - int(rng.integers(0, min(n - 1, 60)))] = ord("N")
        for k in (21, 19):
            want = orc.kmer_hash_sample(s, k, 3)
            got = ctx.kmer_hash_sample(s, k, 3)
            assert got.size == want.size and (got == want).all(), (k, dirty)


@pytest.mark.parametrize("n", [3035, 3036, 3037, 3047, 3048, 3049, 3062, 3063, 3064, 3067, 3068, 3069, 6095, 6096, 6097,
                               27323, 27324, 27325, 27351, 27352, 27431, 27432, 27433, 27452, 27453, 27459, 27460,
                               54647, 54648, 54649, 54675, 54676, 54863, 54864, 54865, 54884, 54885])
def test_hash_sets_shared_image_edges(ctx, orc, n):
    """lengths around the tile / work-item edges of the shared-image kernel (k <= 21: tiles of 3 048 starts, items of
    27 432; k >= 22: 3 036 and 27 324), +- k - 1, with and without a non-base near the end, both strands' images"""
    rng = np.random.default_rng(n)
    s = rand_seq(rng, n)
    for dirty in (False, True):
        if dirty:
            s[n - 1 - int(rng.integers(0, 60))] = ord("N")
        for k, canon in ((21, True), (28, True), (32, True), (22, True), (17, True), (21, False)):
            want = orc.kmer_hash_sample(s, k, 3, canonical=canon)
            got = ctx.kmer_hash_sample(s, k, 3, canonical=canon)
            assert got.size == want.size and (got == want).all(), (k, canon, dirty)


def test_hash_set_scaled1_every_kmer(ctx, orc):
    rng = np.random.default_rng(5)
    s = rand_seq(rng, 30000)
    want = orc.kmer_hash_sample(s, 21, 1)
    got = ctx.kmer_hash_sample(s, 21, 1)
    assert want.size > 29000 and (got == want).all()


def test_non_bases_and_u2t(ctx, orc, hg):
    rng = np.random.default_rng(6)
    s = rand_seq(rng, 40000)
    junk = np.frombuffer(b"NnRYKMSWBDHV-*. \t0\xff\x00Uu", np.uint8)
    pos = rng.choice(40000, 600, replace=False)
    s[pos] = rng.choice(junk, 600)
    for norm in (hg.NORM_ACGT, hg.NORM_U2T):
        want = orc.kmer_hash_sample(s, 21, 5, norm=norm)
        got = ctx.kmer_hash_sample(s, 21, 5, norm=norm)
        assert (got == want).all() and got.size == want.size, norm
    rna = rand_seq(rng, 20000)
    rna[rna == ord("T")] = ord("U")
    a = ctx.kmer_hash_sample(rna, 21, 5, norm=hg.NORM_U2T)
    assert a.size > 1000 and (a == orc.kmer_hash_sample(rna, 21, 5, norm=orc.NORM_U2T)).all()


def test_duplicates_and_low_complexity(ctx, orc):
    # poly-A: one k-mer repeated at every position, all consecutive lanes hit at once
    s = np.full(10000, ord("A"), np.uint8)
    thr = 2**64 - 1
    got = ctx.kmer_hash_sample(s, 21, threshold=thr)
    want = orc.kmer_hash_sample(s, 21, threshold=thr)
    assert want.size == 1 and (got == want).all()
    rng = np.random.default_rng(8)
    rep = np.tile(rand_seq(rng, 997), 40)
    got = ctx.kmer_hash_sample(rep, 21, 2)
    want = orc.kmer_hash_sample(rep, 21, 2)
    assert (got == want).all() and got.size == want.size


def test_capacity_protocol(ctx, hg):
    rng = np.random.default_rng(9)
    s = rand_seq(rng, 5000)
    full = ctx.kmer_hash_sample(s, 21, 1)
    import ctypes as C
    out = np.zeros(10, np.uint64)
    n = C.c_size_t(0)
    st = hg.lib().hg_kmer_hash_sample(ctx._h, s.ctypes.data, s.size, 21, C.c_uint64(2**64 - 1), C.c_uint64(123),
                                      1, 0, out.ctypes.data, 10, C.byref(n))
    assert st == hg.ERR_CAPACITY and n.value == full.size
    st = hg.lib().hg_kmer_hash_sample(ctx._h, s.ctypes.data, s.size, 256, C.c_uint64(1), C.c_uint64(123),
                                      1, 0, out.ctypes.data, 10, C.byref(n))
    assert st == hg.ERR_UNSUPPORTED


def test_synthetic_genome_hash_set(ctx, orc):
    g = orc.synth_genome(3, 600_000)
    want = orc.kmer_hash_sample(g, 21, 1500)
    got = ctx.kmer_hash_sample(g, 21, 1500)
    assert 300 < want.size < 500 and (got == want).all()


# ---- HV encode -----------------------------------------------------------------------------------
@pytest.mark.parametrize("n", [0, 1, 3, 15, 16, 17, 50, 129, 1000, 3333])
@pytest.mark.parametrize("d", [256, 4096])
def test_hv_encode(ctx, orc, hg, n, d):
    rng = np.random.default_rng(n * 7 + d)
    hs = np.unique(rng.integers(0, 2**63, n, dtype=np.uint64))
    for layout, ol in ((hg.LAYOUT_SCALAR, orc.LAYOUT_SCALAR), (hg.LAYOUT_AVX2, orc.LAYOUT_AVX2)):
        hv, n2 = ctx.hv_encode(hs, d, layout)
        want = orc.encode_hv(hs, d, ol)
        assert (hv == want).all(), (n, d, layout)
        assert n2 == orc.hv_norm2(want)


@pytest.mark.parametrize("d", [64, 100, 1000, 1024, 8192, 16384])
def test_hv_encode_other_dims(ctx, orc, hg, d):
    rng = np.random.default_rng(d)
    hs = np.unique(rng.integers(0, 2**63, 300, dtype=np.uint64))
    hv, n2 = ctx.hv_encode(hs, d, hg.LAYOUT_AVX2)
    want = orc.encode_hv(hs, d, orc.LAYOUT_AVX2)
    assert (hv == want).all() and n2 == orc.hv_norm2(want)


@pytest.mark.parametrize("d,layout", [(4096, "avx2"), (8192, "avx2"), (1024, "avx2"), (4096, "scalar"), (1000, "avx2")])
def test_wave_encode_every_plane_count_in_a_large_batch(ctx, orc, hg, d, layout):
    """A batch of >= 8 192 genomes sends every hash set of up to HG_ENC_WAVE_MAX hashes through the one-wave-per-genome
    encoder, whose expansion is specialised on the set size (< 16 through LDS, < 64, < 256, larger): short sequences at
    scaled = 1 (every k-mer is a hash) give sets of 0 .. ~2 900 hashes; each size class is compared with the oracle."""
    rng = np.random.default_rng(d + len(layout))
    n = 8300
    lens = rng.integers(0, 60, n)  # most sets are tiny (0 .. 40 hashes)
    lens[::9] = rng.integers(21, 120, lens[::9].size)
    lens[::37] = rng.integers(200, 400, lens[::37].size)
    lens[::211] = rng.integers(400, 3000, lens[::211].size)
    lens[:6] = (0, 20, 21, 36, 37, 300)  # 0, 0, 1, 16, 17 hashes
    seqs = [rand_seq(rng, int(L)) for L in lens]
    lay, olay = (hg.LAYOUT_AVX2, orc.LAYOUT_AVX2) if layout == "avx2" else (hg.LAYOUT_SCALAR, orc.LAYOUT_SCALAR)
    p = hg.default_params(scaled=1, hv_d=d, hv_layout=lay)
    hv, n2, nh = ctx.sketch_batch(seqs, p)
    classes = {0: 0, 1: 0, 2: 0, 3: 0}
    checked = 0
    for i in list(range(40)) + list(range(40, n, 23)) + list(range(0, n, 211)):
        w_hv, w_n2, w_nh = orc.sketch_genome(seqs[i], scaled=1, hv_d=d, layout=olay)
        assert nh[i] == w_nh and n2[i] == w_n2 and (hv[i] == w_hv).all(), (i, int(lens[i]), w_nh)
        classes[0 if w_nh < 16 else 1 if w_nh < 64 else 2 if w_nh < 256 else 3] += 1
        checked += 1
    assert all(v >= 5 for v in classes.values()), classes
    # every row: the norm is that of the row written, and a set of nh hashes gives entries of nh's parity within +-nh
    hv32 = hv.astype(np.int64)
    assert ((hv32 * hv32).sum(axis=1).astype(np.int32) == n2).all()
    whole = d // 64 * 64
    assert (np.abs(hv32[:, :whole]).max(axis=1) <= nh).all() and ((hv32[:, :whole] - nh[:, None]) % 2 == 0).all()


def test_hv_encode_large_set_wraps_like_i16(ctx, orc, hg):
    rng = np.random.default_rng(11)
    hs = np.unique(rng.integers(0, 2**63, 40000, dtype=np.uint64))  # n > 32767: -(n as i16) wraps
    hv, n2 = ctx.hv_encode(hs, 256, hg.LAYOUT_SCALAR)
    want = orc.encode_hv(hs, 256, orc.LAYOUT_SCALAR)
    assert (hv == want).all() and n2 == orc.hv_norm2(want)


# ---- whole sketch --------------------------------------------------------------------------------
def test_sketch_batch_matches_oracle(ctx, orc, hg):
    rng = np.random.default_rng(12)
    seqs = [orc.synth_genome(g, L) for g, L in ((0, 200_000), (1, 150_001), (57, 99_999), (100, 20), (101, 0), (102, 21))]
    seqs.append(np.concatenate([rand_seq(rng, 5000), np.frombuffer(b"N", np.uint8), rand_seq(rng, 30)]))
    seqs[4] = np.zeros(0, np.uint8)
    p = hg.default_params(scaled=200)
    hv, n2, nh = ctx.sketch_batch(seqs, p)
    for i, s in enumerate(seqs):
        w_hv, w_n2, w_nh = orc.sketch_genome(s, scaled=200)
        assert nh[i] == w_nh and n2[i] == w_n2 and (hv[i] == w_hv).all(), i
    # scalar layout + different parameters
    p2 = hg.default_params(scaled=50, ksize=15, hv_d=1024, hv_layout=hg.LAYOUT_SCALAR, seed=7, canonical=0)
    hv, n2, nh = ctx.sketch_batch(seqs[:3], p2)
    for i in range(3):
        w = orc.sketch_genome(seqs[i], ksize=15, scaled=50, seed=7, canonical=False, hv_d=1024,
                              layout=orc.LAYOUT_SCALAR)
        assert nh[i] == w[2] and n2[i] == w[1] and (hv[i] == w[0]).all()


@pytest.mark.parametrize("scaled,L", [(50, 600_000), (20, 600_000), (5, 600_000), (1, 150_000), (100, 2_000_000)])
def test_dense_sketches_match_oracle(ctx, orc, hg, scaled, L):
    """Sampling rates far above the default (1 k-mer in 5 .. 100 instead of 1 in 1 500): a work item's LDS hit list is
    emptied between tiles or overflows into wave-aggregated appends, and hash sets of 9 000 .. 150 000 keys take the
    value-bucketed sort (LDS-privatised count and scatter, a counting sort per bucket) instead of the one-workgroup one.
    A tandem repeat piles duplicates into single buckets."""
    seqs = [orc.synth_genome(300 + g, L + 17 * g).copy() for g in range(6)]
    unit = seqs[1][1000:1171].copy()
    seqs[1][50_000:50_000 + 171 * 200] = np.tile(unit, 200)  # 200 copies of a 171-base unit
    seqs[2][10_000:10_050] = ord("N")
    p = hg.default_params(scaled=scaled)
    hv, n2, nh = ctx.sketch_batch(seqs, p)
    for i, sq in enumerate(seqs):
        w_hv, w_n2, w_nh = orc.sketch_genome(sq, scaled=scaled)
        assert nh[i] == w_nh and n2[i] == w_n2 and (hv[i] == w_hv).all(), (scaled, i, int(nh[i]), w_nh)
    assert nh.min() > 8192  # every set is past the one-workgroup sort
    for i in (0, 1):
        got = ctx.kmer_hash_sample(seqs[i], 21, scaled)
        want = orc.kmer_hash_sample(seqs[i], 21, scaled)
        assert got.size == want.size and (got == want).all(), (scaled, i)


def test_config1_test_fna(ctx, orc, hg):
    g = golden("g1_test_fna.json")
    seq = orc.read_merge_seq(g["fasta"].encode())
    hv, n2, nh = ctx.sketch_batch([seq])
    assert nh[0] == 0 and n2[0] == 0 and not hv.any()  # empty set at scaled=1500


def test_sketch_5mbp_genome(ctx, orc, hg):
    g = orc.synth_genome(7, 5_000_000)
    hv, n2, nh = ctx.sketch_batch([g])
    w_hv, w_n2, w_nh = orc.sketch_genome(g)
    assert 3100 < w_nh < 3600 and nh[0] == w_nh and n2[0] == w_n2 and (hv[0] == w_hv).all()


def test_large_hit_set_uses_bucketed_sort(ctx, orc):
    # > 16384 sampled hashes in one genome: exceeds the one-workgroup LDS sort -> bucketed multi-workgroup sort
    g = orc.synth_genome(9, 120_000)
    want = orc.kmer_hash_sample(g, 21, 2)
    got = ctx.kmer_hash_sample(g, 21, 2)
    assert want.size > 40000 and (got == want).all()
    # a million distinct hashes (256+ buckets), scaled = 1 (threshold = u64::MAX) and a non-power-of-two count
    g = orc.synth_genome(10, 3_000_000)
    for scaled in (3, 1):
        want = orc.kmer_hash_sample(g[: 1_200_000 if scaled == 1 else g.size], 21, scaled)
        got = ctx.kmer_hash_sample(g[: 1_200_000 if scaled == 1 else g.size], 21, scaled)
        assert want.size > 900_000 and got.size == want.size and (got == want).all(), scaled


def test_large_hit_set_with_heavy_duplicates(ctx, orc):
    """Repeats make raw hit counts far larger than the distinct set: buckets overflow LDS and go through the
    LDS hash-set path; a genome that is one repeated k-mer collapses to a single hash."""
    unit = orc.synth_genome(11, 30_000)[1:]            # 30 kbp unit repeated 40 times: 1.2 M raw hits, ~30 k distinct
    g = np.concatenate([np.frombuffer(b"N", np.uint8)] + [unit] * 40)
    want = orc.kmer_hash_sample(g, 21, 1)
    got = ctx.kmer_hash_sample(g, 21, 1)
    assert 29_000 < want.size < 61_000 and got.size == want.size and (got == want).all()
    poly = np.frombuffer(b"N" + b"A" * 400_000, np.uint8)
    want = orc.kmer_hash_sample(poly, 21, 1)
    got = ctx.kmer_hash_sample(poly, 21, 1)
    assert want.size == 1 and (got == want).all()
    # every distinct key of a 70 k-key genome in ONE hot bucket is impossible for real hashes, but many distinct
    # keys per bucket are not: 50 copies of a 300 kbp unit -> buckets of ~37 k raw / ~750 distinct keys
    unit = orc.synth_genome(12, 300_000)[1:]
    g = np.concatenate([np.frombuffer(b"N", np.uint8)] + [unit] * 12)
    want = orc.kmer_hash_sample(g, 21, 1)
    got = ctx.kmer_hash_sample(g, 21, 1)
    assert got.size == want.size and (got == want).all()


def test_large_hit_set_hash_set_and_inplace_fallback(ctx, orc):
    import os
    # hot hashes (a 40-base unit repeated 50 000 times) on top of a random megabase: their buckets overflow LDS
    # with more than one distinct key -> LDS hash-set de-duplication
    rnd = orc.synth_genome(13, 1_000_000)
    g = np.concatenate([rnd, np.frombuffer(b"N", np.uint8), np.tile(orc.synth_genome(14, 40)[1:], 50_000)])
    want = orc.kmer_hash_sample(g, 21, 1)
    got = ctx.kmer_hash_sample(g, 21, 1)
    assert got.size == want.size and (got == want).all()
    # two buckets for ~100 k distinct keys: the hash set gives up, the genome is sorted in place instead
    g = orc.synth_genome(15, 200_000)
    want = orc.kmer_hash_sample(g, 21, 2)
    try:
        ctx.set_debug("sort_test_buckets", "2")
        got = ctx.kmer_hash_sample(g, 21, 2)
    finally:
        ctx.set_debug("sort_test_buckets", "0")
    assert want.size > 90_000 and got.size == want.size and (got == want).all()


def test_sketch_batch_mixing_small_and_large_sets(ctx, orc, hg):
    # one batch: ordinary genomes (LDS sort) next to one whose set needs the bucketed sort
    p = hg.default_params(scaled=40)
    gs = [orc.synth_genome(20, 200_000), orc.synth_genome(21, 3_000_000), orc.synth_genome(22, 50_000)]
    hv, n2, nh = ctx.sketch_batch(gs, p)
    for i, g in enumerate(gs):
        w_hv, w_n2, w_nh = orc.sketch_genome(g, scaled=40)
        assert nh[i] == w_nh and n2[i] == w_n2 and (hv[i] == w_hv).all(), i
    assert nh[1] > 60_000 and nh[0] < 16_384


# ---- dist ------------------------------------------------------------------------------------------
def _hvs(orc, n_rows, n_hash, d, seed, related=0.6):
    """HVs of hash sets that share a common base set (so that ANIs are high and varied)."""
    rng = np.random.default_rng(seed)
    base = np.unique(rng.integers(0, 2**63, n_hash, dtype=np.uint64))
    rows = []
    for i in range(n_rows):
        keep = base[rng.random(base.size) >= (1.0 - related) * (i % 5) / 4]
        extra = np.unique(rng.integers(0, 2**63, 1 + int(n_hash * 0.05 * (i % 7)), dtype=np.uint64))
        rows.append(orc.encode_hv(np.unique(np.concatenate([keep, extra])), d, orc.LAYOUT_AVX2))
    hv = np.stack(rows)
    n2 = np.array([orc.hv_norm2(r) for r in hv], np.int32)
    return hv, n2


@pytest.mark.parametrize("R,Q,d", [(1, 1, 4096), (5, 7, 4096), (130, 129, 1024), (257, 64, 256), (33, 300, 4096)])
def test_dist_full_matches_oracle(ctx, orc, R, Q, d):
    r, rn = _hvs(orc, R, 400, d, R * 31 + d, related=0.6)
    q, qn = _hvs(orc, Q, 400, d, R * 31 + d, related=0.6)  # same base set => high ANIs
    got = ctx.dist_full(r, rn, q, qn, 21)
    want = orc.ani_matrix(r, rn, q, qn, 21)
    assert np.abs(got - want).max() <= ANI_TOL
    assert (want > 80).any() and (got[want == 0] == 0).all()


def test_dist_exact_for_large_values(ctx, orc):
    # |hv| beyond f16's exact range and norms beyond the f32-exact window: still bit-exact dots
    rng = np.random.default_rng(21)
    r = rng.integers(-3000, 3000, (40, 1024)).astype(np.int16)
    q = np.vstack([r[:10], rng.integers(-3000, 3000, (20, 1024)).astype(np.int16)])
    rn = np.array([orc.hv_norm2(x) for x in r], np.int32)
    qn = np.array([orc.hv_norm2(x) for x in q], np.int32)
    want = orc.ani_matrix(r, rn, q, qn, 21)
    assert np.abs(ctx.dist_full(r, rn, q, qn, 21) - want).max() <= ANI_TOL
    # thresholded entry point: the speculative whole-K launch must veto itself and the integer kernel decide
    hits = ctx.dist(r, rn, q, qn, 21, symmetric=False, ani_th=50.0)
    assert {(int(h["ref_idx"]), int(h["qry_idx"])) for h in hits} == {(i, j) for i, j in zip(*np.nonzero(want >= 50.0))}
    assert len(hits) >= 10 and all(abs(h["ani"] - want[h["ref_idx"], h["qry_idx"]]) <= ANI_TOL for h in hits)
    # mid-size values: f16-exact but needing chunked accumulation
    r = rng.integers(-1500, 1500, (40, 4096)).astype(np.int16)
    q = np.vstack([r[:10] + rng.integers(-20, 20, (10, 4096)).astype(np.int16), r[10:25]])
    rn = np.array([orc.hv_norm2(x) for x in r], np.int32)
    qn = np.array([orc.hv_norm2(x) for x in q], np.int32)
    got, want = ctx.dist_full(r, rn, q, qn, 21), orc.ani_matrix(r, rn, q, qn, 21)
    assert np.abs(got - want).max() <= ANI_TOL and (want > 99).any()


def test_dist_thresholded_and_symmetric(ctx, orc, hg):
    r, rn = _hvs(orc, 150, 500, 4096, 77, related=0.5)
    full = orc.ani_matrix(r, rn, r, rn, 21)
    hits = ctx.dist(r, rn, r, rn, 21, symmetric=True, ani_th=85.0)
    want = {(i, j) for i in range(150) for j in range(i + 1, 150) if full[i, j] >= 85.0}
    assert {(int(h["ref_idx"]), int(h["qry_idx"])) for h in hits} == want and len(hits) == len(want)
    for h in hits:
        assert abs(h["ani"] - full[h["ref_idx"], h["qry_idx"]]) <= ANI_TOL
    allp = ctx.dist(r, rn, r[:40], rn[:40], 21, symmetric=False, ani_th=90.0)
    assert len(allp) == int((full[:, :40] >= 90.0).sum())
    srt = hg.sort_ani_hits(allp, 40)
    assert (np.diff(srt["ani"]) <= 0).all()


def test_dist_of_sketched_synthetic_cluster(ctx, orc, hg):
    seqs = [orc.synth_genome(g, 300_000) for g in (0, 10, 50, 99, 100)]
    hv, n2, nh = ctx.sketch_batch(seqs, hg.default_params(scaled=100))
    ani = ctx.dist_full(hv, n2, hv, n2, 21)
    want = orc.ani_matrix(hv, n2, hv, n2, 21)
    assert np.abs(ani - want).max() <= ANI_TOL
    assert ani[0, 0] == 100.0 and 98.5 < ani[0, 1] < 99.5 and 94 < ani[0, 2] < 96 and ani[0, 4] < 85


@pytest.mark.parametrize("R,Q,d,sym", [(3000, 1, 4096, False), (3000, 10, 4096, False), (5, 2500, 4096, False), (16, 700, 1024, False),
                                       (700, 16, 8192, False), (9, 9, 4096, True), (2000, 3, 256, False), (1, 1, 4096, False),
                                       (50, 3, 1000, False), (40, 2, 8, False), (33, 7, 4104, False), (7, 1001, 520, False)])
def test_dist_with_a_handful_of_rows_on_one_side(ctx, orc, hg, R, Q, d, sym):
    """Up to 16 rows on one side (one or a few genomes against a database): the streaming kernel -- the small side in LDS,
    exact int32 dot products by v_dot2_i32_i16 -- must report the oracle's pairs and ANIs, on either side and symmetric."""
    import torch
    rng = np.random.default_rng(R * 31 + Q * 7 + d)
    n = 2000
    base = (rng.integers(0, 2, (24, d)) * 2 - 1).astype(np.int64)
    def sketches(m):
        hv = np.zeros((m, d), np.int64)
        for i in range(m):
            hv[i] = -n + 2 * rng.binomial(n, 0.5, d) + 300 * base[i % 24]  # members of 24 families: planted similar pairs
        return hv.astype(np.int16)
    r = sketches(R)
    q = r if sym else sketches(Q)
    rn, qn = (r.astype(np.int64) ** 2).sum(1).astype(np.int32), (q.astype(np.int64) ** 2).sum(1).astype(np.int32)
    want = orc.ani_matrix(r, rn, q, qn, 21)
    dev = torch.device("cuda:0")
    tr, tq = torch.from_numpy(r).to(dev), torch.from_numpy(q).to(dev)
    trn, tqn = torch.from_numpy(rn).to(dev), torch.from_numpy(qn).to(dev)
    th = 60.0
    cap = R * Q
    hits = torch.empty(cap * 3, dtype=torch.int32, device=dev)
    found, st = ctx.dist_dev(tr.data_ptr(), trn.data_ptr(), R, (tr if sym else tq).data_ptr(), (trn if sym else tqn).data_ptr(), Q, d, 21, sym, th,
                             hits.data_ptr(), cap)
    assert st == 0 and ctx.last_kernel("dist").startswith("dist_skinny_kernel") == (min(R, Q) * d * 2 <= 128 * 1024)  # (the small side fits LDS)
    h = hits.cpu().numpy()[: 3 * found].reshape(-1, 3)
    got = {(int(a), int(b)): float(np.array([c], np.int32).view(np.float32)[0]) for a, b, c in h}
    exp = {(i, j) for i in range(R) for j in range(Q) if want[i, j] >= th and (not sym or i < j)}
    assert set(got) == exp and len(got) == found  # the oracle's set: no tolerance band around the threshold
    assert all(abs(v - want[k]) <= ANI_TOL for k, v in got.items())
    assert len(exp) > 0 or R * Q < 4


# ---- bit-packed extension (BASELINE configs[4]; the oracle's popcount definition is the reference) ----
@pytest.mark.parametrize("R,Q,d", [(3000, 1, 2048), (3000, 10, 16384), (700, 16, 1024), (700, 17, 1024), (3000, 32, 2048), (300, 33, 2048)])
def test_hamming_search_with_a_handful_of_queries(ctx, orc, R, Q, d):
    """Searches with up to 16 / 32 queries take the popcount kernel's narrow tiles (128 references x 16 / 32 queries), whatever
    the number of references; the full matrix and the thresholded hits must be the oracle's."""
    import torch
    rng = np.random.default_rng(R + 7 * Q + d)
    wr = rng.integers(0, 2**32, (R, d // 32), dtype=np.uint64).astype(np.uint32)
    wq = wr[rng.choice(R, Q, replace=False)].copy()
    wq[:, 0] ^= rng.integers(0, 2**32, Q, dtype=np.uint64).astype(np.uint32)  # near copies: planted hits
    if Q > 2:
        wq[Q // 2:] = rng.integers(0, 2**32, (Q - Q // 2, d // 32), dtype=np.uint64).astype(np.uint32)
    dev = torch.device("cuda:0")
    br, bq = torch.from_numpy(wr.view(np.int32)).to(dev), torch.from_numpy(wq.view(np.int32)).to(dev)
    want = orc.hamming_matrix(wr, wq)
    out = torch.empty((R, Q), dtype=torch.int32, device=dev)
    ctx.hamming_full_dev(br.data_ptr(), R, bq.data_ptr(), Q, d, out.data_ptr())
    ctx.sync()
    assert (out.cpu().numpy().view(np.uint32) == want).all()
    max_dist = int(d * 0.47)
    cap = R * Q
    hits = torch.empty(cap * 3, dtype=torch.int32, device=dev)
    n, st = ctx.hamming_search_dev(br.data_ptr(), R, bq.data_ptr(), Q, d, max_dist, hits.data_ptr(), cap)
    assert st == 0 and ctx.last_hamming_path() == (0 if Q <= 32 else ctx.last_hamming_path())
    exp = {(i, j, int(want[i, j])) for i, j in zip(*np.nonzero(want <= max_dist))}
    got = hits.cpu().numpy().view(np.uint32)[: 3 * n].reshape(-1, 3)
    assert n == len(exp) >= min(Q, 2) // 2 and {(int(a), int(b), int(c)) for a, b, c in got} == exp


@pytest.mark.parametrize("R,Q,d", [(1, 1, 128), (37, 300, 16384), (200, 129, 4096), (130, 5, 256), (70, 330, 2464), (9, 600, 96)])
def test_binarize_and_hamming(ctx, orc, R, Q, d):
    import torch
    rng = np.random.default_rng(R * 1000 + Q)
    r = rng.integers(-40, 40, (R, d)).astype(np.int16)
    q = np.vstack([r[: min(R, Q) // 2], rng.integers(-40, 40, (Q - min(R, Q) // 2, d)).astype(np.int16)])
    dev = torch.device("cuda:0")
    tr, tq = torch.from_numpy(r).to(dev), torch.from_numpy(q).to(dev)
    br = torch.empty((R, d // 32), dtype=torch.int32, device=dev)
    bq = torch.empty((Q, d // 32), dtype=torch.int32, device=dev)
    ctx.hv_binarize_dev(tr.data_ptr(), R, d, br.data_ptr())
    ctx.hv_binarize_dev(tq.data_ptr(), Q, d, bq.data_ptr())
    ctx.sync()
    wr, wq = orc.binarize(r), orc.binarize(q)
    assert (br.cpu().numpy().view(np.uint32) == wr).all() and (bq.cpu().numpy().view(np.uint32) == wq).all()
    want = orc.hamming_matrix(wr, wq)
    if d % 128 == 0:  # (the xor + popcount kernel reads rows in 16-byte pieces)
        out = torch.empty((R, Q), dtype=torch.int32, device=dev)
        ctx.hamming_full_dev(br.data_ptr(), R, bq.data_ptr(), Q, d, out.data_ptr())
        ctx.sync()
        assert (out.cpu().numpy().view(np.uint32) == want).all()
    # thresholded search: exactly the pairs at distance <= max_dist
    max_dist = int(np.percentile(want, 30))
    cap = R * Q
    hits = torch.empty(cap * 3, dtype=torch.int32, device=dev)
    exp = {(i, j, int(want[i, j])) for i, j in zip(*np.nonzero(want <= max_dist))}
    # xor + popcount kernel / +-1.0 e2m1 GEMM (v_mfma_scale_f32_16x16x128_f8f6f4) / +-1 byte GEMM (v_mfma_i32_16x16x64_i8, D % 128 == 0
    # only): the same integers.  D = 2464 and 96 are not whole K-steps: the FP4 path pads with zero nibbles.
    paths = {"popc": 0, "fp4": 2, "mfma": 1} if d % 128 == 0 else {"fp4": 2, "": 2}
    for path, code in paths.items():
        for tile in ("big", "wide"):
            ctx.set_debug("ham_path", path)
            ctx.set_debug("dist_tile", tile)
            try:
                n, st = ctx.hamming_search_dev(br.data_ptr(), R, bq.data_ptr(), Q, d, max_dist, hits.data_ptr(), cap)
            finally:
                ctx.set_debug("ham_path", "")
                ctx.set_debug("dist_tile", "")
            assert ctx.last_hamming_path() == code
            got = hits.cpu().numpy().view(np.uint32)[: 3 * n].reshape(-1, 3)
            assert st == 0 and n == len(exp), (path, tile)
            assert {(int(a), int(b), int(c)) for a, b, c in got} == exp, (path, tile)
    # every pair is a hit when max_dist >= D, none of the padded rows / columns leaks in
    for path in ("fp4",) + (("mfma",) if d % 128 == 0 else ()):
        ctx.set_debug("ham_path", path)
        try:
            n, st = ctx.hamming_search_dev(br.data_ptr(), R, bq.data_ptr(), Q, d, d, hits.data_ptr(), cap)
        finally:
            ctx.set_debug("ham_path", "")
        got = hits.cpu().numpy().view(np.uint32)[: 3 * n].reshape(-1, 3)
        assert st == 0 and n == R * Q and {(int(a), int(b), int(c)) for a, b, c in got} == {(i, j, int(want[i, j])) for i in range(R) for j in range(Q)}


def test_hamming_of_d16384_sketches(ctx, orc, hg):
    import torch
    seqs = [orc.synth_genome(g, 200_000) for g in (0, 20, 60, 100)]
    hv, n2, nh = ctx.sketch_batch(seqs, hg.default_params(scaled=100, hv_d=16384))
    dev = torch.device("cuda:0")
    t = torch.from_numpy(hv).to(dev)
    b = torch.empty((4, 512), dtype=torch.int32, device=dev)
    ctx.hv_binarize_dev(t.data_ptr(), 4, 16384, b.data_ptr())
    out = torch.empty((4, 4), dtype=torch.int32, device=dev)
    ctx.hamming_full_dev(b.data_ptr(), 4, b.data_ptr(), 4, 16384, out.data_ptr())
    ctx.sync()
    d = out.cpu().numpy()
    assert (d == orc.hamming_matrix(orc.binarize(hv), orc.binarize(hv))).all()
    assert d[0, 0] == 0 and d[0, 1] < d[0, 2] < d[0, 3] and 7000 < d[0, 3] < 9400  # unrelated ~ D/2


# ---- robustness -------------------------------------------------------------------------------------
def test_hit_buffer_overflow_retry(ctx, orc, hg):
    """A sampled k-mer repeated thousands of times overflows the expected-size hit buffer: the raw counter
    keeps counting, the batch is re-run with exact capacities, results stay exact (lossless, unlike the
    reference's 8 slots per thread, src/cuda_kernel.cu:316)."""
    rng = np.random.default_rng(77)
    thr = (2**64 - 1) // 1500
    kmer = None
    for _ in range(20000):
        cand = rand_seq(rng, 21)
        if orc.kmer_hash_sample(cand, 21, threshold=thr).size == 1:
            kmer = cand
            break
    assert kmer is not None
    unit = np.concatenate([kmer, np.frombuffer(b"N", np.uint8)])
    seq = np.concatenate([np.tile(unit, 4000), rand_seq(rng, 100_000)])  # 4000 raw hits, cap would be ~1.3 k
    other = orc.synth_genome(5, 150_000)
    hv, n2, nh = ctx.sketch_batch([other, seq, other])
    for i, s in enumerate([other, seq, other]):
        w = orc.sketch_genome(s)
        assert nh[i] == w[2] and n2[i] == w[1] and (hv[i] == w[0]).all(), i
    got = ctx.kmer_hash_sample(seq, 21, 1500)
    assert (got == orc.kmer_hash_sample(seq, 21, 1500)).all()


def test_dev_api_argument_checks(ctx, hg):
    import torch
    dev = torch.device("cuda:0")
    seq = torch.zeros(4096, dtype=torch.uint8, device=dev)
    hv = torch.empty((2, 4096), dtype=torch.int16, device=dev)
    n2 = torch.empty(2, dtype=torch.int32, device=dev)
    nh = torch.empty(2, dtype=torch.int32, device=dev)
    p = hg.default_params()
    with pytest.raises(hg.HgError) as e:  # offsets must be multiples of 4
        ctx.sketch_batch_dev(seq.data_ptr(), [0, 1001], [1000, 1000], p, hv.data_ptr(), n2.data_ptr(), nh.data_ptr())
    assert e.value.status == hg.ERR_INVALID
    for bad in (dict(ksize=0), dict(ksize=256), dict(scaled=0), dict(hv_d=0), dict(hv_layout=7)):
        with pytest.raises(hg.HgError):
            ctx.sketch_batch_dev(seq.data_ptr(), [0, 1000], [1000, 1000], hg.default_params(**bad), hv.data_ptr(),
                                 n2.data_ptr(), nh.data_ptr())
    ctx.sketch_batch_dev(seq.data_ptr(), [0, 1000], [1000, 1000], p, hv.data_ptr(), n2.data_ptr(), nh.data_ptr())
    ctx.sync()
    assert int(nh.sum()) == 0  # all-zero bytes are not bases


def test_dist_big_tiles_and_chunked_paths(ctx, orc, hg):
    """Large thresholded problems take the 256 x 256 geometry; rows whose norms exceed the f32-exact
    window take the chunked (i32 side accumulator) variant -- both against the oracle."""
    rng = np.random.default_rng(31)
    R, Q, D = 700, 1300, 4096
    base = rng.integers(-60, 60, (8, D))
    r = (base[rng.integers(0, 8, R)] + rng.integers(-25, 25, (R, D))).astype(np.int16)
    q = (base[rng.integers(0, 8, Q)] + rng.integers(-25, 25, (Q, D))).astype(np.int16)
    rn = np.array([orc.hv_norm2(x) for x in r], np.int32)
    qn = np.array([orc.hv_norm2(x) for x in q], np.int32)
    want = orc.ani_matrix(r, rn, q, qn, 21)
    th = float(np.float32(np.percentile(want, 90)))  # (a float32 value: the library compares in float32)
    import os
    key = lambda h: np.sort(h, order=["ref_idx", "qry_idx"])
    try:
        ctx.set_debug("dist_tile", "big")       # 256 x 256, LDS-DMA staging (swizzled image)
        hits_big = ctx.dist(r, rn, q, qn, 21, symmetric=False, ani_th=th)
        ctx.set_debug("dist_tile", "wide")      # 256 x 320, LDS-DMA staging
        hits_wide = ctx.dist(r, rn, q, qn, 21, symmetric=False, ani_th=th)
        ctx.set_debug("dist_tile", "small")     # 128 x 128
        hits = ctx.dist(r, rn, q, qn, 21, symmetric=False, ani_th=th)
    finally:
        ctx.set_debug("dist_tile", "")
    assert (key(hits_big) == key(hits)).all()      # all geometries: identical hits, bit for bit
    assert (key(hits_wide) == key(hits)).all()
    sel = want >= th
    # (with ANI_TOL = 0 no pair is "near": the hit set is the oracle's)
    near = np.abs(want - th) <= ANI_TOL
    got = {(int(h["ref_idx"]), int(h["qry_idx"])) for h in hits}
    must = {(i, j) for i, j in zip(*np.nonzero(sel & ~near))}
    may = {(i, j) for i, j in zip(*np.nonzero(sel | near))}
    assert must <= got <= may
    for h in hits[:2000]:
        assert abs(h["ani"] - want[h["ref_idx"], h["qry_idx"]]) <= ANI_TOL
    # big values: |x| up to ~900 -> norms ~1e9: exact only through chunked accumulation
    r2 = (r.astype(np.int32) * 12).astype(np.int16)
    q2 = (q.astype(np.int32) * 12).astype(np.int16)
    rn2 = np.array([orc.hv_norm2(x) for x in r2], np.int32)
    qn2 = np.array([orc.hv_norm2(x) for x in q2], np.int32)
    want2 = orc.ani_matrix(r2[:300], rn2[:300], q2[:400], qn2[:400], 21)
    got2 = ctx.dist_full(r2[:300], rn2[:300], q2[:400], qn2[:400], 21)
    assert np.abs(got2 - want2).max() <= ANI_TOL
    hits2 = ctx.dist(r2, rn2, q2, qn2, 21, ani_th=float(np.percentile(want2, 50)))
    full2 = orc.ani_matrix(r2, rn2, q2, qn2, 21)
    for h in hits2[:3000]:
        assert abs(h["ani"] - full2[h["ref_idx"], h["qry_idx"]]) <= ANI_TOL


def test_symmetric_large(ctx, orc, hg):
    rng = np.random.default_rng(41)
    n, D = 900, 4096
    base = rng.integers(-60, 60, (6, D))
    r = (base[rng.integers(0, 6, n)] + rng.integers(-30, 30, (n, D))).astype(np.int16)
    rn = np.array([orc.hv_norm2(x) for x in r], np.int32)
    want = orc.ani_matrix(r, rn, r, rn, 21)
    th = float(np.float32(np.percentile(want, 80) + 3e-4))  # (a float32 value: the library compares in float32)
    hits = ctx.dist(r, rn, r, rn, 21, symmetric=True, ani_th=th)
    near = np.abs(want - th) <= ANI_TOL
    iu = np.triu(np.ones((n, n), bool), 1)
    got = {(int(h["ref_idx"]), int(h["qry_idx"])) for h in hits}
    assert {(i, j) for i, j in zip(*np.nonzero((want >= th) & iu & ~near))} <= got
    assert got <= {(i, j) for i, j in zip(*np.nonzero(((want >= th) | near) & iu))}


def test_sketch_batch_of_many_small_genomes_is_packed_for_upload(ctx, orc, hg):
    # >= 16 genomes of < 1 MiB each in a sub-batch take the packed (one upload) path of hg_sketch_batch
    rng = np.random.default_rng(77)
    p = hg.default_params(scaled=50)
    gs = [rand_seq(rng, int(n)) for n in rng.integers(0, 30_000, 200)]
    gs[7] = gs[7][:5]      # shorter than k
    gs[8] = gs[8][:0]      # empty
    for rep in range(2):   # the second call reuses the pinned staging buffers
        hv, n2, nh = ctx.sketch_batch(gs, p)
        for i in (0, 1, 7, 8, 9, 57, 199):
            w_hv, w_n2, w_nh = orc.sketch_genome(gs[i], scaled=50)
            assert nh[i] == w_nh and n2[i] == w_n2 and (hv[i] == w_hv).all(), (rep, i)


def test_tiny_hash_sets_sorted_by_one_wave(ctx, orc, hg):
    """Hash sets of 2..64 raw hits are ordered and de-duplicated by one wave in registers (no LDS network): every count in
    that range, the counts just above it (65..70: the workgroup's network again), duplicates -- a unit repeated so that the
    same hashes come several times --, and the sets' HVs; checked for EVERY genome of the batch"""
    rng = np.random.default_rng(4242)
    p = hg.default_params(scaled=30)
    gs = []
    for n in list(range(700, 2400, 17)) + [100, 21, 22, 40]:  # ~1 hash per 30 bases: 1 .. 80 raw hits
        g = rand_seq(rng, n)
        if n % 3 == 0:  # a third of the genomes: their first 300 bases three more times (duplicate hashes)
            g = np.concatenate([g, g[:300], g[:300], g[:300]])
        gs.append(g)
    hv, n2, nh = ctx.sketch_batch(gs, p)
    sizes = set()
    for i, g in enumerate(gs):
        want = orc.kmer_hash_sample(g, 21, 30)
        got = ctx.kmer_hash_sample(g, 21, 30)
        assert got.size == want.size and (got == want).all(), i
        w_hv, w_n2, w_nh = orc.sketch_genome(g, scaled=30)
        assert nh[i] == w_nh and n2[i] == w_n2 and (hv[i] == w_hv).all(), i
        sizes.add(int(w_nh))
    assert min(sizes) <= 2 and max(sizes) >= 66 and len(sizes) >= 40


def test_hash_sets_with_repeats_take_the_sort_fallback(ctx, orc):
    """The LDS sort's counting-sort fast path buckets the sampled hashes by value (0.8 keys per bucket for uniform
    hashes); equal hashes share a bucket, so a genome with a segment repeated 40 times piles 40 keys into ~100 buckets:
    past the per-bucket limit the genome must take the bitonic network after all.  Unique genomes of the same batch stay
    on the fast path."""
    rng = np.random.default_rng(777)
    unit = rand_seq(rng, 2000)
    rep = np.concatenate([rand_seq(rng, 150_000)] + [unit] * 40 + [rand_seq(rng, 50_000)])
    mild = np.concatenate([rand_seq(rng, 200_000), unit, unit, unit])   # three copies: inside the limit
    uniq = rand_seq(rng, 260_000)
    for s in (rep, mild, uniq):
        for scaled in (20, 60):
            want = orc.kmer_hash_sample(s, 21, scaled)
            got = ctx.kmer_hash_sample(s, 21, scaled)
            assert want.size >= 512 and got.size == want.size and (got == want).all(), scaled
    import hypergen_amd as hg
    p = hg.default_params(scaled=20)
    hv, n2, nh = ctx.sketch_batch([rep, mild, uniq, rep], p)
    for i, g in enumerate((rep, mild, uniq, rep)):
        w_hv, w_n2, w_nh = orc.sketch_genome(g, scaled=20)
        assert nh[i] == w_nh and n2[i] == w_n2 and (hv[i] == w_hv).all(), i


def test_repeated_batch_geometry_with_growing_hash_sets(ctx, orc, hg):
    """A batch with the geometry of the previous one reuses its plan, and its LDS sort is sized by the COUNTS the
    previous run saw: genomes that come back with many more sampled hashes (same lengths, fewer non-bases) must be
    picked up by the capacity-sized sort again -- and a third run, sized by the larger counts, agrees as well."""
    rng = np.random.default_rng(31337)
    p = hg.default_params(scaled=200)
    full = [rand_seq(rng, int(n)) for n in (300_000, 120_000, 500_000, 64_000)]
    sparse = []
    for g in full:
        h = g.copy()
        h[rng.random(h.size) < 0.12] = ord("N")  # a non-base every ~8 bases: few 21-mers survive
        sparse.append(h)
    for gs in (sparse, full, full, sparse):
        hv, n2, nh = ctx.sketch_batch(gs, p)
        for i in range(len(gs)):
            w_hv, w_n2, w_nh = orc.sketch_genome(gs[i], scaled=200)
            assert nh[i] == w_nh and n2[i] == w_n2 and (hv[i] == w_hv).all(), i
    assert nh.max() * 8 < ctx.sketch_batch(full, p)[2].max()  # (the sparse sets are indeed much smaller)


def _sketch_like(rng, rows, D, n, base=None, share=0.6):
    """HVs with the structure of real sketches: hv = 2*count - n (uniform parity per row), count ~ Binomial."""
    cnt = rng.binomial(n, 0.5, (rows, D))
    if base is not None:
        k = int(n * share)
        cnt = base[rng.integers(0, base.shape[0], rows)] + rng.binomial(n - k, 0.5, (rows, D))
    return (2 * cnt - n).astype(np.int16)


@pytest.mark.parametrize("same", [True, False])
def test_dist_i8_operand_path_equals_f16_and_oracle(ctx, orc, same):
    """The i8 operand path (centred counts on v_mfma_i32_16x16x64_i8, outlier entries as extra K columns, parity /
    row-sum correction in the epilogue) must give the integers of the f16 path: identical hits, identical floats.
    Rows carry planted outliers (|count - n/2| > 127) -- alone in their dimension, shared by two rows, and at the
    same dimension on both sides -- odd and even n, a ragged D."""
    rng = np.random.default_rng(77 if same else 78)
    D = 4096 - 8 * 5
    n_odd, n_even = 3333, 3000
    base = rng.binomial(int(n_odd * 0.6), 0.5, (6, D))
    R, Q = 900, 900 if same else 700
    r = _sketch_like(rng, R, D, n_odd, base)
    r[::3] = _sketch_like(rng, len(r[::3]), D, n_even, rng.binomial(int(n_even * 0.6), 0.5, (6, D)))
    q = r if same else _sketch_like(rng, Q, D, n_odd, base)
    # planted outliers: +-(300..480) keeps |x| <= 509 (residual fits a byte) and the row parity
    def plant(m, row, d, v):
        m[row, d] = v if (v - m[row, 0]) % 2 == 0 else v + 1
    plant(r, 5, 17, 401), plant(r, 5, 3000, -455), plant(r, 77, 17, 377), plant(r, 400, 4000, 480), plant(r, 899, 0, -300)
    if not same:
        plant(q, 9, 17, -391), plant(q, 9, 2222, 333), plant(q, 650, 4000, 445)
    rn = np.array([orc.hv_norm2(x) for x in r], np.int32)
    qn = rn if same else np.array([orc.hv_norm2(x) for x in q], np.int32)
    want = orc.ani_matrix(r, rn, q, qn, 21)
    key = lambda h: np.sort(h, order=["ref_idx", "qry_idx"])
    try:
        for sym in ((False, True) if same else (False,)):
            ctx.set_debug("dist_path", "f16")
            h16 = key(ctx.dist(r, rn, q, qn, 21, symmetric=sym, ani_th=60.0))
            ctx.set_debug("dist_path", "i8")
            h8 = key(ctx.dist(r, rn, q, qn, 21, symmetric=sym, ani_th=60.0))
            assert h8.size == h16.size > 1000 and np.array_equal(h8, h16), (same, sym)
            got = np.zeros_like(want)
            got[h8["ref_idx"], h8["qry_idx"]] = h8["ani"]
            sel = want >= 60.0 + ANI_TOL
            if sym:
                sel = np.triu(sel, 1)
            assert (got[sel] > 0).all() and np.abs(got[sel] - want[sel]).max() <= ANI_TOL
            # pairs through the planted outliers, including the diagonal of an outlier row
            for (i, j) in ((5, 77), (5, 5), (400, 400), (899, 3)) if same else ((5, 9), (400, 650), (77, 9)):
                if want[i, j] >= 60.0 and (not sym or i < j):
                    assert abs(got[i, j] - want[i, j]) <= ANI_TOL, (i, j)
    finally:
        ctx.set_debug("dist_path", "")


@pytest.mark.parametrize("path", ["i8", "f16"])
def test_dist_tile_orders_and_epilogue_paths_agree(ctx, orc, path):
    """A self-comparison runs its diagonal tiles first (hg_ctx_set_debug "dist_order" = "plain" turns that off; "legacy"
    = the kernel's own blockIdx -> tile mapping instead of the host's balanced slot table): the
    same hits every way, symmetric or not, on a size with ragged last tiles.  The HVs are clustered in blocks of 150 so
    that diagonal tiles carry dense blocks (candidate lists that overflow in the middle of a tile: the slab path with
    its cooperative flushes) while the others see scattered candidates or none (the lane-mask path, one list per tile,
    and the early exit); a low and a high threshold move tiles between the three."""
    rng = np.random.default_rng(4242)
    D, n, R = 4096, 3333, 1900
    base = rng.binomial(int(n * 0.8), 0.5, (R // 150 + 1, D))
    cnt = base[np.arange(R) // 150] + rng.binomial(n - int(n * 0.8), 0.5, (R, D))
    hv = (2 * cnt - n).astype(np.int16)
    rn = np.array([orc.hv_norm2(x) for x in hv], np.int32)
    key = lambda h: np.sort(h, order=["ref_idx", "qry_idx"])
    want = orc.ani_matrix(hv[:400], rn[:400], hv, rn, 21)
    try:
        ctx.set_debug("dist_path", path)
        for th in (80.0, 95.0):
            for sym in (False, True):
                ctx.set_debug("dist_order", "plain")
                a = key(ctx.dist(hv, rn, hv, rn, 21, symmetric=sym, ani_th=th))
                ctx.set_debug("dist_order", "legacy")  # the workgroups derive their tiles from blockIdx (no host table)
                l = key(ctx.dist(hv, rn, hv, rn, 21, symmetric=sym, ani_th=th))
                ctx.set_debug("dist_order", "")
                b = key(ctx.dist(hv, rn, hv, rn, 21, symmetric=sym, ani_th=th))
                assert a.size == b.size > 10000 and np.array_equal(a, b) and np.array_equal(l, b), (th, sym)
                got = np.zeros((400, R), np.float32)
                m = b["ref_idx"] < 400
                got[b["ref_idx"][m], b["qry_idx"][m]] = b["ani"][m]
                sel = want >= th + ANI_TOL
                if sym:
                    sel &= np.arange(400)[:, None] < np.arange(R)[None, :]
                assert (got[sel] > 0).all() and np.abs(got[sel] - want[sel]).max() <= ANI_TOL, (th, sym)
                assert not (got[want < th - ANI_TOL] > 0).any()
    finally:
        ctx.set_debug("dist_path", "")
        ctx.set_debug("dist_order", "")


@pytest.mark.parametrize("path", ["i8", "f16"])
def test_dist_every_pair_passes_on_ragged_corner_tiles(ctx, orc, path):
    """ani_th <= 0 makes the pre-filter's column bound -inf; the corner tile of a ragged matrix has a few valid rows and
    columns next to many padded ones, few enough candidates for the lane-mask path -- the padded rows must stay out
    (their "out of range" threshold used to cancel against the -inf: garbage hits beyond R)."""
    rng = np.random.default_rng(515)
    D, n = 4096, 3001
    for R, Q in ((300, 325), (260, 330), (129, 131)):
        r = (2 * rng.binomial(n, 0.5, (R, D)) - n).astype(np.int16)
        q = (2 * rng.binomial(n, 0.5, (Q, D)) - n).astype(np.int16)
        rn = np.array([orc.hv_norm2(x) for x in r], np.int32)
        qn = np.array([orc.hv_norm2(x) for x in q], np.int32)
        try:
            ctx.set_debug("dist_path", path)
            for tile in ("wide", "big", ""):
                ctx.set_debug("dist_tile", tile)
                for th in (0.0, -3.0):
                    h = ctx.dist(r, rn, q, qn, 21, symmetric=False, ani_th=th)
                    assert h.size == R * Q, (R, Q, tile, th, h.size)
                    assert h["ref_idx"].max() < R and h["qry_idx"].max() < Q
                    assert np.unique(h["ref_idx"].astype(np.int64) * Q + h["qry_idx"]).size == R * Q
        finally:
            ctx.set_debug("dist_path", "")
            ctx.set_debug("dist_tile", "")


@pytest.mark.parametrize("seed", range(6))
def test_dist_random_shapes_thresholded_equals_full(ctx, orc, seed):
    """Random shapes, dimensions, thresholds and cluster structure, both operand formats: the thresholded hit list must
    be exactly the full matrix cut at the threshold (ragged last tiles, empty tiles, tiles with a few candidates and
    tiles that are one dense block meet in one launch)."""
    rng = np.random.default_rng(9000 + seed)
    D = int(rng.choice([512, 1000, 2048, 4096, 4096 - 24]))
    n = int(rng.integers(600, 3400))
    R, Q = int(rng.integers(1, 1400)), int(rng.integers(1, 1400))
    nb = int(rng.integers(1, 12))
    base = rng.binomial(int(n * 0.75), 0.5, (nb, D))
    def make(rows):
        cnt = base[rng.integers(0, nb, rows)] + rng.binomial(n - int(n * 0.75), 0.5, (rows, D))
        return (2 * cnt - n).astype(np.int16)
    same = bool(rng.integers(0, 2))
    r = make(R)
    q = r if same else make(Q)
    rn = np.array([orc.hv_norm2(x) for x in r], np.int32)
    qn = rn if same else np.array([orc.hv_norm2(x) for x in q], np.int32)
    full = ctx.dist_full(r, rn, q, qn, 21)
    key = lambda h: np.sort(h, order=["ref_idx", "qry_idx"])
    try:
        for path in ("i8", "f16"):
            ctx.set_debug("dist_path", path)
            for th in (float(rng.uniform(60, 90)), 99.5):
                for sym in ((False, True) if same else (False,)):
                    h = key(ctx.dist(r, rn, q, qn, 21, symmetric=sym, ani_th=th))
                    sel = full >= th
                    if sym:
                        sel &= np.arange(r.shape[0])[:, None] < np.arange(q.shape[0])[None, :]
                    wi, wj = np.nonzero(sel)
                    assert h.size == wi.size, (seed, path, th, sym, h.size, wi.size)
                    assert np.array_equal(h["ref_idx"], wi) and np.array_equal(h["qry_idx"], wj)
                    assert np.array_equal(h["ani"], full[wi, wj]), (seed, path, th, sym)
    finally:
        ctx.set_debug("dist_path", "")


@pytest.mark.parametrize("n,expect_i8", [(4400, True), (5200, True), (9000, False)])
def test_dist_i8_reach_mid_size_sketches(ctx, orc, n, expect_i8):
    """Sketches of 4 000-5 500 hashes have a few clamped entries in most rows (|count - n/2| > 127 at ~3.5 sigma): every
    row's entries are one contiguous range of the entry list, both operands' entries meet in the same dimension now and
    then -- the i8 path must take them (automatically) and give the f16 path's hits bit for bit; at 9 000 hashes the
    list overflows its two-entries-per-row capacity and the call falls back to f16 by itself."""
    rng = np.random.default_rng(n)
    D, R, Q = 4096, 640, 500
    base = rng.binomial(int(n * 0.6), 0.5, (5, D))
    r, q = _sketch_like(rng, R, D, n, base), _sketch_like(rng, Q, D, n, base)
    rn = np.array([orc.hv_norm2(x) for x in r], np.int32)
    qn = np.array([orc.hv_norm2(x) for x in q], np.int32)
    clamped = int((np.abs((r.astype(np.int32) + (r[:, :1] & 1)) >> 1) > 127).sum())
    assert (clamped > 200) if expect_i8 else (clamped > 2 * (R + Q) + 1024)
    key = lambda h: np.sort(h, order=["ref_idx", "qry_idx"])
    want = orc.ani_matrix(r, rn, q, qn, 21)
    try:
        ctx.set_debug("dist_path", "i8")  # (small problem: the automatic choice starts at 256^3 pairs x dims)
        h8 = key(ctx.dist(r, rn, q, qn, 21, symmetric=False, ani_th=60.0))
        assert ctx.last_dist_path() == (1 if expect_i8 else 0)
        ctx.set_debug("dist_path", "f16")
        h16 = key(ctx.dist(r, rn, q, qn, 21, symmetric=False, ani_th=60.0))
    finally:
        ctx.set_debug("dist_path", "")
    assert h8.size == h16.size > 1000 and np.array_equal(h8, h16)
    got = np.zeros_like(want)
    got[h8["ref_idx"], h8["qry_idx"]] = h8["ani"]
    sel = want >= 60.0 + ANI_TOL
    assert (got[sel] > 0).all() and np.abs(got[sel] - want[sel]).max() <= ANI_TOL


@pytest.mark.parametrize("n,expect", [(7000, 3), (12000, 3), (15000, 3), (20000, 0)])
def test_dist_centred_f16_operands_for_large_sketches(ctx, orc, n, expect):
    """Sketches beyond the reach of byte operands (~6 000 hashes at D = 4096): the centred counts (x + e) >> 1 as f16 are
    exact in ONE f32 accumulation window up to ~15 500 hashes (max row sum c^2 ~ n D / 4 <= 2^24), so they take the whole-K kernel
    (path 3) -- same hits, bit for bit, as the windowed raw-value path and the oracle's ANI; at 20 000 hashes the bound
    fails on the device and the raw-value chain queued behind runs (path 0).  A row of mixed parity has no centred form."""
    rng = np.random.default_rng(n)
    D, R, Q = 4096, 640, 500
    base = rng.binomial(int(n * 0.6), 0.5, (5, D))
    r, q = _sketch_like(rng, R, D, n, base), _sketch_like(rng, Q, D, n, base)
    rn = np.array([orc.hv_norm2(x) for x in r], np.int32)
    qn = np.array([orc.hv_norm2(x) for x in q], np.int32)
    key = lambda h: np.sort(h, order=["ref_idx", "qry_idx"])
    want = orc.ani_matrix(r, rn, q, qn, 21)
    try:
        ctx.set_debug("dist_path", "cen")  # (small problem: the automatic choice starts at 256^3 pairs x dims)
        hc = key(ctx.dist(r, rn, q, qn, 21, symmetric=False, ani_th=60.0))
        assert ctx.last_dist_path() == expect
        if expect == 3:
            assert ctx.last_kernel("dist").endswith(", true>")  # dist_mfma_kernel<..., CEN = true>
            hc2 = key(ctx.dist(r, rn, q, qn, 21, symmetric=False, ani_th=60.0))  # (the repeat: raw chain not queued)
            assert ctx.last_dist_path() == 3 and np.array_equal(hc, hc2)
            bad = r.copy()
            bad[17, 100] += 1  # mixed parity
            bn = np.array([orc.hv_norm2(x) for x in bad], np.int32)
            hb = key(ctx.dist(bad, bn, q, qn, 21, symmetric=False, ani_th=60.0))
            assert ctx.last_dist_path() == 0
            ctx.set_debug("dist_path", "f16")
            assert np.array_equal(hb, key(ctx.dist(bad, bn, q, qn, 21, symmetric=False, ani_th=60.0)))
        ctx.set_debug("dist_path", "f16")
        h16 = key(ctx.dist(r, rn, q, qn, 21, symmetric=False, ani_th=60.0))
        assert ctx.last_dist_path() == 0
    finally:
        ctx.set_debug("dist_path", "")
    assert hc.size == h16.size > 1000 and np.array_equal(hc, h16)
    got = np.zeros_like(want)
    got[hc["ref_idx"], hc["qry_idx"]] = hc["ani"]
    sel = want >= 60.0 + ANI_TOL
    assert (got[sel] > 0).all() and np.abs(got[sel] - want[sel]).max() <= ANI_TOL


def test_dist_i8_vetoed_inputs_fall_back_to_f16(ctx, orc):
    """Inputs the i8 path must refuse on the device -- a row of mixed parity, a residual beyond a byte, more outlier
    dims than extra columns -- still give the f16 result (the f16 kernels queued behind the attempt run)."""
    rng = np.random.default_rng(79)
    D, R = 4096, 700
    base = rng.binomial(2000, 0.5, (5, D))
    for kind in ("clean", "mixed_parity", "clean", "huge_value", "too_many_outliers"):
        # ("clean" in between: a successful i8 call makes the next call on the same buffers trust the i8 path and
        # queue no f16 fallback -- the vetoed input right after it must then be rerun through the f16 schedule)
        r = _sketch_like(rng, R, D, 3333, base)
        if kind == "clean":
            pass
        elif kind == "mixed_parity":
            r[123, 77] += 1
        elif kind == "huge_value":
            r[5, 9] = 1201 if r[5, 0] % 2 else 1200
        else:
            r[:, ::3] = (r[:, ::3].astype(np.int32) * 3 + (r[:, :1].astype(np.int32) % 2) * -2).astype(np.int16)  # sigma x3, parity kept
        rn = np.array([orc.hv_norm2(x) for x in r], np.int32)
        want = orc.ani_matrix(r, rn, r, rn, 21)
        ctx.set_debug("dist_path", "i8")
        try:
            h = ctx.dist(r, rn, r, rn, 21, symmetric=True, ani_th=70.0)
        finally:
            ctx.set_debug("dist_path", "")
        got = np.zeros_like(want)
        got[h["ref_idx"], h["qry_idx"]] = h["ani"]
        sel = np.triu(want >= 70.0 + ANI_TOL, 1)
        assert h.size >= int(sel.sum()) > 100 and np.abs(got[sel] - want[sel]).max() <= ANI_TOL, kind
        assert ctx.last_dist_path() == (1 if kind == "clean" else 0), kind


# ---- independent golden vectors straight through the kernels ---------------------------------------
def test_wyrng_golden_streams_through_the_encode_kernel(hg, ctx):
    """tests/golden/kat_wyrng.json (tools/gen_golden_wyrng.py: the published wyrng definition in Python integers) against
    the DEVICE's WyRng: the scalar-layout HV of a one-hash set is hv[64 i + j] = -1 + 2 * bit j of word i
    (src/hd.rs:100-107), so the encode kernels hand back the generator's words bit for bit."""
    g = golden("kat_wyrng.json")
    for st in g["streams"]:
        want = [int(x, 16) for x in st["next_u64"]]
        hv, n2 = ctx.hv_encode(np.array([int(st["seed"], 16)], np.uint64), 64 * len(want), hg.LAYOUT_SCALAR)
        assert set(np.unique(hv).tolist()) <= {-1, 1} and n2 == hv.size
        bits = ((hv.astype(np.int32) + 1) >> 1).reshape(len(want), 64)
        got = [sum(int(b) << j for j, b in enumerate(row)) for row in bits]
        assert got == want, st["seed"]
        # the AVX2 order of the same words: hv[64 i + 4 (j & 15) + (j >> 4)] (src/hd.rs:84-87)
        hv2, _ = ctx.hv_encode(np.array([int(st["seed"], 16)], np.uint64), 64 * len(want), hg.LAYOUT_AVX2)
        j = np.arange(64)
        assert np.array_equal(hv2.reshape(-1, 64)[:, 4 * (j & 15) + (j >> 4)], hv.reshape(-1, 64))
    # two hashes in one set: counts add (word-wise popcount of the two streams)
    a, b = g["streams"][0], g["streams"][3]
    hv, _ = ctx.hv_encode(np.array(sorted([int(a["seed"], 16), int(b["seed"], 16)]), np.uint64), 1024, hg.LAYOUT_SCALAR)
    wa, wb = [int(x, 16) for x in a["next_u64"]], [int(x, 16) for x in b["next_u64"]]
    want = np.array([[-2 + 2 * (((x >> j) & 1) + ((y >> j) & 1)) for j in range(64)] for x, y in zip(wa, wb)]).ravel()
    assert np.array_equal(hv, want)


def _vectors_with_dot(dot, d=64):
    """two i16 vectors of dimension d whose i32 dot product is exactly `dot` (|dot| <= 2^31)"""
    s, m = (-1 if dot < 0 else 1), abs(dot)
    big = 32767 * 32767
    c, rem = divmod(m, big)
    a, b = divmod(rem, 32767)
    r, q = np.zeros(d, np.int64), np.zeros(d, np.int64)
    for i in range(c):
        r[i], q[i] = 32767, 32767
    r[c], q[c] = a, 32767
    r[c + 1], q[c + 1] = b, 1
    assert int(r @ q) == m and c + 2 <= d
    return (s * r).astype(np.int16), q.astype(np.int16)


def test_ani_golden_through_the_dist_kernels(hg, ctx, orc):
    """tests/golden/g4_ani.json (tools/gen_golden_cpu.py: src/dist.rs:153-160 in numpy float32 with the i32-wrapping
    denominator; not the oracle) through hg_dist_full and hg_dist: row i / column i carry vectors whose dot product is
    case i's `dot` and the case's norms -- dot = nr = nq, den = 0, negative dots, wrapped denominators, wrapped norms."""
    cases = golden("g4_ani.json")
    n = len(cases)
    r = np.zeros((n, 64), np.int16)
    q = np.zeros((n, 64), np.int16)
    for i, c in enumerate(cases):
        r[i], q[i] = _vectors_with_dot(c["dot"])
    rn = np.array([c["nr"] for c in cases], np.int64).astype(np.int32)
    qn = np.array([c["nq"] for c in cases], np.int64).astype(np.int32)
    for k in sorted({c["k"] for c in cases}):
        sel = [i for i, c in enumerate(cases) if c["k"] == k]
        full = ctx.dist_full(r, rn, q, qn, k)
        for i in sel:
            assert abs(float(full[i, i]) - cases[i]["ani"]) <= ANI_TOL, cases[i]
        want = orc.ani_matrix(r, rn, q, qn, k)  # off-diagonal pairs: the oracle
        assert float(np.abs(full - want).max()) <= ANI_TOL
        for th in (0.0, 50.0, 96.0):
            hits = ctx.dist(r, rn, q, qn, k, symmetric=False, ani_th=th)
            got = {(int(h["ref_idx"]), int(h["qry_idx"])): float(h["ani"]) for h in hits}
            sure = {(i, j) for i in range(n) for j in range(n) if want[i, j] >= th + ANI_TOL}
            maybe = {(i, j) for i in range(n) for j in range(n) if want[i, j] >= th - ANI_TOL}
            assert sure <= set(got) <= maybe, (k, th)
            assert all(abs(v - want[i, j]) <= ANI_TOL for (i, j), v in got.items())
            for i in sel:
                if cases[i]["ani"] >= th + ANI_TOL:
                    assert abs(got[(i, i)] - cases[i]["ani"]) <= ANI_TOL


def test_hamming_tile_orders_agree(ctx, orc):
    """a search of several tiles a side (6 x 5 at 256 x 256, 6 x 5 at 256 x 320): the host's slot -> tile table and the
    kernel's own blockIdx mapping report the same pairs, and they are the oracle's"""
    import torch
    rng = np.random.default_rng(99)
    R, Q, d = 1500, 1300, 1024
    r = rng.integers(-40, 40, (R, d)).astype(np.int16)
    q = np.vstack([r[:400] + (rng.random((400, d)) < 0.02) * 90, rng.integers(-40, 40, (Q - 400, d))]).astype(np.int16)
    dev = torch.device("cuda:0")
    wr, wq = orc.binarize(r), orc.binarize(q)
    br, bq = torch.from_numpy(wr.view(np.int32)).to(dev), torch.from_numpy(wq.view(np.int32)).to(dev)
    want = orc.hamming_matrix(wr, wq)
    max_dist = 300
    exp = {(i, j, int(want[i, j])) for i, j in zip(*np.nonzero(want <= max_dist))}
    assert 300 < len(exp) < 5000
    hits = torch.empty(3 * 20000, dtype=torch.int32, device=dev)
    try:
        for order in ("", "legacy"):
            for tile in ("big", "wide"):
                ctx.set_debug("dist_order", order)
                ctx.set_debug("dist_tile", tile)
                n, st = ctx.hamming_search_dev(br.data_ptr(), R, bq.data_ptr(), Q, d, max_dist, hits.data_ptr(), 20000)
                got = hits.cpu().numpy().view(np.uint32)[: 3 * n].reshape(-1, 3)
                assert st == 0 and {(int(a), int(b), int(c)) for a, b, c in got} == exp, (order, tile)
    finally:
        ctx.set_debug("dist_order", "")
        ctx.set_debug("dist_tile", "")


@pytest.mark.parametrize("symmetric", [False, True])
def test_comparisons_beyond_the_hit_counter_run_as_row_blocks(hg, orc, symmetric):
    """The kernels count hits in 32 bits, so a comparison of more than 2^32 - 1 pairs (66 000 x 66 000 and up) runs as blocks
    of reference rows.  The hook "pair_limit" lowers that border so that the path can be checked at a size that fits a test:
    same hit set as the one-launch call (global indices, the i < j rule), the count added up over the blocks, and the
    capacity contract -- a buffer that fills up half way still gets the total count back with HG_ERR_CAPACITY."""
    import torch
    import bench
    dev = torch.device("cuda:0")
    n = 3000
    hv = bench.clustered_hvs(n, 0, dev, n=1200)
    n2 = (hv.int() ** 2).sum(1).int()
    qv = hv if symmetric else bench.clustered_hvs(n - 400, 0, dev, n=1200, salt=1)
    qn = n2 if symmetric else (qv.int() ** 2).sum(1).int()
    Q = qv.shape[0]
    cap = 2_000_000
    out = torch.empty(cap * 3, dtype=torch.int32, device=dev)

    def canon(t, k):
        h = t[: 3 * k].view(-1, 3).cpu().numpy().view(np.uint32)
        return h[np.lexsort((h[:, 1], h[:, 0]))]
    with hg.Context(0) as c:
        want_n, st = c.dist_dev(hv.data_ptr(), n2.data_ptr(), n, qv.data_ptr(), qn.data_ptr(), Q, 4096, 21, symmetric, 80.0, out.data_ptr(), cap)
        assert st == 0 and want_n > 50_000
        want = canon(out, want_n)
        for limit in (n * Q // 3 + 17, 5 * Q + 1, Q):  # three blocks; blocks of five rows; one row per launch
            if limit < 6 * Q and symmetric:
                continue  # (600+ launches: once is enough)
            c.set_debug("pair_limit", str(limit))
            out.zero_()
            torch.cuda.synchronize()  # (torch's stream, not the ctx's own: the fill must not overtake the first block's hits)
            got_n, st = c.dist_dev(hv.data_ptr(), n2.data_ptr(), n, qv.data_ptr(), qn.data_ptr(), Q, 4096, 21, symmetric, 80.0, out.data_ptr(), cap)
            assert st == 0 and got_n == want_n, limit
            assert np.array_equal(canon(out, got_n), want), limit
        # the buffer fills up inside the second block: everything that fits is a hit of the full list, the count is the total
        c.set_debug("pair_limit", str(n * Q // 3 + 17))
        small = want_n // 2
        out.zero_()
        torch.cuda.synchronize()
        got_n, st = c.dist_dev(hv.data_ptr(), n2.data_ptr(), n, qv.data_ptr(), qn.data_ptr(), Q, 4096, 21, symmetric, 80.0, out.data_ptr(), small)
        assert st == hg.ERR_CAPACITY and got_n == want_n
        part = canon(out, small)
        full = {(int(a), int(b)): int(v) for a, b, v in want}
        assert all(full.get((int(a), int(b))) == int(v) for a, b, v in part[:: max(1, small // 5000)])
        if not symmetric:  # ... also when the blocks are a handful of rows (the streaming kernel): the trailing ones only count
            c.set_debug("pair_limit", str(5 * Q + 1))
            out.zero_()
            torch.cuda.synchronize()
            got_n, st = c.dist_dev(hv.data_ptr(), n2.data_ptr(), n, qv.data_ptr(), qn.data_ptr(), Q, 4096, 21, symmetric, 80.0, out.data_ptr(), small)
            assert st == hg.ERR_CAPACITY and got_n == want_n and c.last_kernel("dist").startswith("dist_skinny_kernel")
            part = canon(out, small)
            assert all(full.get((int(a), int(b))) == int(v) for a, b, v in part[:: max(1, small // 5000)])
        c.set_debug("pair_limit", "0")
        # the bit-packed search takes the same route
        bits = torch.empty((n, 128), dtype=torch.int32, device=dev)
        c.hv_binarize_dev(hv.data_ptr(), n, 4096, bits.data_ptr())
        hn, st = c.hamming_search_dev(bits.data_ptr(), n, bits.data_ptr(), n, 4096, 1200, out.data_ptr(), cap)
        assert st == 0 and hn >= n
        hw = canon(out, hn)
        c.set_debug("pair_limit", str(n * n // 4 + 5))
        out.zero_()
        torch.cuda.synchronize()
        hn2, st = c.hamming_search_dev(bits.data_ptr(), n, bits.data_ptr(), n, 4096, 1200, out.data_ptr(), cap)
        assert st == 0 and hn2 == hn and np.array_equal(canon(out, hn2), hw)
        c.set_debug("pair_limit", "0")
