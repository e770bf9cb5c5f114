"""Seeded random parity sweep of the HIP path against the CPU oracle: random lengths (incl. 0 and < k), mixed-case
and non-ACGT bytes at random densities, every kernel family (k 1..8, 9..21, 22..32, 33..255), random `scaled`
incl. 1, both strands modes and both normalisation modes; then ragged batches through hg_sketch_batch and random
thresholded dist calls against the full ANI matrix."""
import numpy as np
import pytest

from conftest import ANI_TOL

pytestmark = pytest.mark.gpu

import os

ALPHA = np.frombuffer(b"ACGTacgtNnRYKMUu-.*\n\r>", np.uint8)
SOAK = int(os.environ.get("HG_FUZZ_SEEDS", "0"))  # extra seeds for a one-off soak run


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


def random_genome(rng, n):
    """ACGT with a random share of lower case and of arbitrary other bytes (runs and singletons)."""
    s = rng.choice(ALPHA[:4], n)
    if n == 0:
        return s
    lower = rng.random(n) < rng.choice([0.0, 0.01, 0.3])
    s = np.where(lower, s | 0x20, s).astype(np.uint8)
    junk = rng.random(n) < rng.choice([0.0, 0.0005, 0.01, 0.08])
    s = np.where(junk, rng.choice(ALPHA[4:], n), s).astype(np.uint8)
    for _ in range(int(rng.integers(0, 4))):  # runs of N, as between contigs / in scaffolds
        a = int(rng.integers(0, n))
        s[a: a + int(rng.integers(1, 300))] = ord("N")
    return s


@pytest.mark.parametrize("seed", range(24 + SOAK))
def test_random_hash_sets(ctx, orc, hg, seed):
    rng = np.random.default_rng(7000 + seed)
    for _ in range(6):
        k = int(rng.choice([rng.integers(1, 9), rng.integers(9, 22), rng.integers(22, 33), rng.integers(33, 80)]))
        n = int(rng.choice([0, k - 1 if k > 1 else 0, k, k + 1, rng.integers(k, 4000), rng.integers(4000, 150_000)]))
        s = random_genome(rng, n)
        scaled = int(rng.choice([1, 2, 7, 50, 1500]))
        if k <= 8 and n > 40_000:
            scaled = max(scaled, 7)  # keep the oracle quick
        canonical = bool(rng.integers(0, 2))
        norm = int(rng.choice([hg.NORM_ACGT, hg.NORM_U2T]))
        sd = int(rng.choice([123, 0, 2**63 + 5]))
        want = orc.kmer_hash_sample(s, k, scaled, sd, canonical, norm)
        got = ctx.kmer_hash_sample(s, k, scaled, sd, canonical, norm)
        assert got.size == want.size and (got == want).all(), (seed, k, n, scaled, canonical, norm, sd)


@pytest.mark.parametrize("seed", range(6 + SOAK // 4))
def test_random_ragged_batches(ctx, orc, hg, seed):
    rng = np.random.default_rng(8000 + seed)
    k = int(rng.choice([15, 21, 27, 31]))
    scaled = int(rng.choice([20, 300, 1500]))
    d = int(rng.choice([256, 1024, 4096]))
    layout = int(rng.choice([hg.LAYOUT_SCALAR, hg.LAYOUT_AVX2]))
    p = hg.default_params(ksize=k, scaled=scaled, hv_d=d, hv_layout=layout, canonical=int(rng.integers(0, 2)))
    lens = [0, k - 1, k, 3 * k] + [int(x) for x in rng.integers(100, 400_000, 9)]
    rng.shuffle(lens)
    gs = [random_genome(rng, n) for n in lens]
    hv, n2, nh = ctx.sketch_batch(gs, p)
    for i, g in enumerate(gs):
        w_hv, w_n2, w_nh = orc.sketch_genome(g, k, scaled, 123, bool(p.canonical), hg.NORM_ACGT, d, layout)
        assert nh[i] == w_nh and n2[i] == w_n2 and (hv[i] == w_hv).all(), (seed, i, lens[i])


@pytest.mark.parametrize("seed", range(6 + SOAK // 4))
def test_random_thresholded_dist(ctx, orc, seed):
    rng = np.random.default_rng(9000 + seed)
    R, Q = int(rng.integers(1, 700)), int(rng.integers(1, 700))
    d = int(rng.choice([256, 1024, 4096]))
    amp = int(rng.choice([40, 300, 1500, 2500]))  # 2500: beyond f16's exact range -> integer kernel
    base = rng.integers(-amp, amp, (6, d))
    r = np.clip(base[rng.integers(0, 6, R)] + rng.integers(-amp // 3, amp // 3 + 1, (R, d)), -32768, 32767).astype(np.int16)
    q = np.clip(base[rng.integers(0, 6, Q)] + rng.integers(-amp // 3, amp // 3 + 1, (Q, d)), -32768, 32767).astype(np.int16)
    rn = np.array([orc.hv_norm2(x) for x in r], np.int32)
    qn = np.array([orc.hv_norm2(x) for x in q], np.int32)
    k = int(rng.choice([16, 21, 31]))
    want = orc.ani_matrix(r, rn, q, qn, k)
    assert np.abs(ctx.dist_full(r, rn, q, qn, k) - want).max() <= ANI_TOL
    for th in (float(np.percentile(want, 30)), float(np.percentile(want, 95)), 0.0, 100.5):
        hits = ctx.dist(r, rn, q, qn, k, symmetric=False, ani_th=th)
        got = {(int(h["ref_idx"]), int(h["qry_idx"])) for h in hits}
        near = np.abs(want - th) <= ANI_TOL
        must = {(i, j) for i, j in zip(*np.nonzero((want >= th) & ~near))}
        may = {(i, j) for i, j in zip(*np.nonzero((want >= th) | near))}
        assert must <= got <= may and len(hits) == len(got), (seed, th)
        for h in hits[:: max(1, len(hits) // 500)]:
            assert abs(h["ani"] - want[h["ref_idx"], h["qry_idx"]]) <= ANI_TOL
