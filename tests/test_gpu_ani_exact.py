"""ANI as a function of the dot product: EQUAL to the oracle's, bit for bit -- not "within 1e-4".

`ani = 1 + ln(2 / (1 / J + 1)) / k` (src/dist.rs:153-160) is float32 arithmetic plus one call of the C library's logf
(Rust's f32::ln).  The divisions and additions are IEEE operations on both sides; the logarithm is glibc's table-driven
routine, which the device evaluates operation for operation (hyper-gen_amd/csrc/hg_logf.h).  The oracle calls the host's
logf itself (oracle/hg_oracle.c: orc_ani_from_dot), so these tests compare with the real thing:
  * every float in (0, 1] -- the whole range 2 / (1 / J + 1) can take -- plus specials and random patterns elsewhere;
  * the formula on integer (dot, norm, norm) tuples, edge cases included;
  * full matrices, thresholded hit SETS and hit order of the GEMM paths.
"""
import numpy as np
import pytest
import torch

from conftest import golden

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def hg():
    import hypergen_amd
    return hypergen_amd


@pytest.fixture(scope="module")
def ctx(hg):
    c = hg.Context(0)
    c.set_stream(torch.cuda.current_stream().cuda_stream)
    yield c
    c.close()


def same_bits(a, b):
    """float arrays equal bit for bit, any NaN equal to any NaN"""
    a, b = np.ascontiguousarray(a, np.float32), np.ascontiguousarray(b, np.float32)
    na, nb = np.isnan(a), np.isnan(b)
    return bool((na == nb).all() and (a.view(np.uint32)[~na] == b.view(np.uint32)[~na]).all())


def test_logf_equals_the_hosts_on_every_float_in_0_1(ctx, orc):
    """1 065 353 217 bit patterns: subnormals, every normal float up to 1.0"""
    CH = 1 << 26
    out = torch.empty(CH, dtype=torch.float32, device="cuda:0")
    last = 0x3F800000
    first = 1
    while first <= last:
        n = min(CH, last - first + 1)
        ctx.logf_dev(out.data_ptr(), n, first_bits=first)
        ctx.sync()
        got = out[:n].cpu().numpy()
        x = np.arange(first, first + n, dtype=np.uint32).view(np.float32)
        want = orc.logf_array(x)
        if not same_bits(got, want):
            bad = np.nonzero(got.view(np.uint32) != want.view(np.uint32))[0]
            raise AssertionError("logf differs at bit pattern 0x%08x: device %r, host %r (%d in this chunk)" % (
                first + int(bad[0]), got[bad[0]], want[bad[0]], bad.size))
        first += n


def test_logf_specials_and_the_rest_of_the_number_line(ctx, orc):
    rng = np.random.default_rng(11)
    bits = np.concatenate([
        np.array([0x00000000, 0x80000000, 0x00000001, 0x007FFFFF, 0x00800000, 0x3F7FFFFF, 0x3F800000, 0x3F800001, 0x7F7FFFFF,
                  0x7F800000, 0xFF800000, 0x7FC00000, 0xFFC00000, 0x7F800001, 0xBF800000, 0x80000001, 0x3F330000, 0x3F32FFFF],
                 np.uint32),
        rng.integers(0, 1 << 32, 1 << 22, dtype=np.uint64).astype(np.uint32),
        rng.integers(0x3F800000, 0x7F800000, 1 << 22, dtype=np.uint64).astype(np.uint32)])
    x = torch.from_numpy(bits.view(np.float32).copy()).cuda()
    out = torch.empty_like(x)
    ctx.logf_dev(out.data_ptr(), x.numel(), d_x=x.data_ptr())
    ctx.sync()
    assert same_bits(out.cpu().numpy(), orc.logf_array(bits.view(np.float32)))


def test_ani_formula_on_integer_tuples(ctx, orc):
    rng = np.random.default_rng(12)
    n = 1 << 22
    nr = rng.integers(1, 40_000_000, n).astype(np.int32)
    nq = rng.integers(1, 40_000_000, n).astype(np.int32)
    dot = (np.sqrt(nr.astype(np.float64) * nq) * rng.random(n) ** 0.3).astype(np.int32)  # 0 .. the Cauchy-Schwarz bound, dense near it
    # edge cases in front: equal sets (J = 1), dot 0 / negative (J <= 0 -> 0), zero norms (0 / 0 -> NaN -> 0), i32 wrap of the
    # denominator, dot > norms
    edge = np.array([[5, 5, 5], [0, 7, 9], [-3, 7, 9], [0, 0, 0], [1, 2**31 - 1, 2**31 - 1], [13_650_000, 13_650_000, 13_650_000],
                     [10, 3, 4], [2**31 - 1, 2**31 - 1, 2**31 - 1], [1, 1, 1], [1, 1, 2], [-2**31, 5, 5]], np.int64).astype(np.int32)
    dot[:len(edge)], nr[:len(edge)], nq[:len(edge)] = edge[:, 0], edge[:, 1], edge[:, 2]
    d = [torch.from_numpy(a).cuda() for a in (dot, nr, nq)]
    for k in (21, 1, 16, 31, 255):
        out = torch.empty(n, dtype=torch.float32, device="cuda:0")
        ctx.ani_from_dots_dev(d[0].data_ptr(), d[1].data_ptr(), d[2].data_ptr(), n, k, out.data_ptr())
        ctx.sync()
        got, want = out.cpu().numpy(), orc.ani_from_dots(dot, nr, nq, k)
        assert same_bits(got, want), (k, np.nonzero(got != want)[0][:5])
        assert (got[[1, 2, 3]] == 0).all() and got[0] == 100.0


def test_ani_golden_tuples_are_exact(ctx):
    """tests/golden/g4_ani.json: tools/gen_golden_cpu.py, glibc's logf algorithm in Python doubles -- shares no code with the
    oracle or the library"""
    cases = golden("g4_ani.json")
    dot, nr, nq = (np.array([c[key] for c in cases], np.int64).astype(np.int32) for key in ("dot", "nr", "nq"))
    d = [torch.from_numpy(a).cuda() for a in (dot, nr, nq)]
    for k in sorted({c["k"] for c in cases}):
        out = torch.empty(len(cases), dtype=torch.float32, device="cuda:0")
        ctx.ani_from_dots_dev(d[0].data_ptr(), d[1].data_ptr(), d[2].data_ptr(), len(cases), k, out.data_ptr())
        ctx.sync()
        got = out.cpu().numpy()
        for i, c in enumerate(cases):
            if c["k"] == k:
                assert np.float32(c["ani"]) == got[i], (c, got[i])


def clustered(orc, n, D=4096, seed=3, nhash=3333, clusters=8):
    """HVs of hash sets that share a graded fraction of their hashes inside a cluster (ANI 85..100 inside, ~0 across)"""
    rng = np.random.default_rng(seed)
    hv, n2 = np.zeros((n, D), np.int16), np.zeros(n, np.int32)
    roots = [rng.integers(0, 1 << 63, nhash, dtype=np.uint64) for _ in range(clusters)]
    for i in range(n):
        root = roots[i % clusters]
        keep = int(nhash * (1.0 - 0.6 * (i // clusters) / max(1, n // clusters)))
        hs = np.unique(np.concatenate([root[:keep], rng.integers(0, 1 << 63, nhash - keep, dtype=np.uint64)]))
        hv[i] = orc.encode_hv(hs, D)
        n2[i] = np.int32((hv[i].astype(np.int64) ** 2).sum() & 0xFFFFFFFF)
    return hv, n2


@pytest.mark.parametrize("path", ["", "i8", "f16", "cen"])
def test_full_matrix_and_hit_set_equal_the_oracle(ctx, orc, path):
    hv, n2 = clustered(orc, 384)
    want = orc.ani_matrix(hv, n2, hv, n2, 21)
    ctx.set_debug("dist_path", path)
    try:
        got = ctx.dist_full(hv, n2, hv, n2, 21)
        assert same_bits(got, want)
        for th in (85.0, 95.0, float(np.sort(want.ravel())[-2000])):  # the last one: a threshold that IS a value of the matrix
            hits = ctx.dist(hv, n2, hv, n2, 21, ani_th=th)
            got_set = {(int(h["ref_idx"]), int(h["qry_idx"])): np.float32(h["ani"]) for h in hits}
            want_set = {(int(i), int(j)): want[i, j] for i, j in zip(*np.nonzero(want >= np.float32(th)))}
            assert got_set.keys() == want_set.keys(), (path, th, len(got_set), len(want_set))
            assert all(got_set[k2] == v for k2, v in want_set.items())
    finally:
        ctx.set_debug("dist_path", "")


def test_sorted_hits_are_in_the_references_order(ctx, orc, hg):
    """dump_ani_file (src/utils.rs:260-308): stable ascending sort by ANI, then reversed -- descending, ties in REVERSE
    enumeration order.  With exact values the device order is the oracle's order, pair for pair."""
    hv, n2 = clustered(orc, 256, seed=5)
    want = orc.ani_matrix(hv, n2, hv, n2, 21)
    hits = ctx.dist(hv, n2, hv, n2, 21, ani_th=80.0)
    srt = hg.sort_ani_hits(hits, 256)
    pairs = [(i, j) for i in range(256) for j in range(256) if want[i, j] >= np.float32(80.0)]  # enumeration order (dist.rs:243-265)
    order = sorted(range(len(pairs)), key=lambda t: want[pairs[t]])  # stable ascending ...
    order.reverse()  # ... then reversed
    assert [(int(h["ref_idx"]), int(h["qry_idx"])) for h in srt] == [pairs[t] for t in order]
    assert all(np.float32(h["ani"]) == want[h["ref_idx"], h["qry_idx"]] for h in srt)
