"""Device-side post-processing of hit lists: dump_ani_file's order (src/utils.rs:262-269) from
two stable radix passes on the device, and top-k per query (the body of the reference's empty `search`)."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu
D = 4096


@pytest.fixture(scope="module")
def hits_10k():
    import bench
    import hypergen_amd as hg
    dev = torch.device("cuda:0")
    ctx = hg.Context(0)
    ctx.set_stream(torch.cuda.current_stream().cuda_stream)
    n = 10_000
    hv = bench.clustered_hvs(n, 0, dev)
    n2 = (hv.int() ** 2).sum(1).int()
    cap = 2_000_000
    hits = torch.empty(cap * 3, dtype=torch.int32, device=dev)
    found, st = ctx.dist_dev(hv.data_ptr(), n2.data_ptr(), n, hv.data_ptr(), n2.data_ptr(), n, D, 21, False, 85.0,
                             hits.data_ptr(), cap)
    torch.cuda.synchronize()
    assert st == 0 and found > 1_000_000  # the 10^6-hit list of BASELINE configs[3]
    yield ctx, hits, found, n
    ctx.close()


def test_device_order_equals_dump_ani_file_order_at_1e6_hits(hits_10k):
    import hypergen_amd as hg
    ctx, hits, found, n = hits_10k
    raw = hits[: 3 * found].cpu().numpy().view(hg.ANI_HIT_DTYPE).copy()
    # duplicates of ANI values exist (self pairs at 100.0, equal floats), so the tie rule is exercised
    assert np.unique(raw["ani"]).size < raw.size
    want = hg.sort_ani_hits(raw, n)  # the host implementation of the reference's order (model-tested on CPU)
    work = hits[: 3 * found].clone()
    ctx.sort_ani_hits_dev(work.data_ptr(), found, n)
    torch.cuda.synchronize()
    got = work.cpu().numpy().view(hg.ANI_HIT_DTYPE)
    assert np.array_equal(got, want)
    assert np.array_equal(ctx.sort_ani_hits_staged(raw, n), want)
    # independent model on a slice: stable ascending by ANI over the enumeration, reversed
    sub = raw[raw["ref_idx"] < 40]
    order = np.lexsort((sub["ref_idx"].astype(np.int64) * n + sub["qry_idx"], sub["ani"]))[::-1]
    assert np.array_equal(hg.sort_ani_hits(sub, n), sub[order])


@pytest.mark.parametrize("k", [1, 3, 150])
def test_topk_per_query(hits_10k, k):
    import hypergen_amd as hg
    ctx, hits, found, n = hits_10k
    dev = hits.device
    out = torch.empty((n * k, 3), dtype=torch.int32, device=dev)
    cnt = torch.empty(n, dtype=torch.int32, device=dev)
    ctx.topk_per_query_dev(hits.data_ptr(), found, n, k, out.data_ptr(), cnt.data_ptr())
    torch.cuda.synchronize()
    raw = hits[: 3 * found].cpu().numpy().view(hg.ANI_HIT_DTYPE)
    got = out.cpu().numpy().view(hg.ANI_HIT_DTYPE).reshape(n, k)
    c = cnt.cpu().numpy()
    per_q = np.bincount(raw["qry_idx"], minlength=n)
    assert np.array_equal(c, np.minimum(per_q, k))
    order = np.lexsort((raw["ref_idx"], -raw["ani"].astype(np.float64), raw["qry_idx"]))  # qry asc, ani desc, ref asc
    s = raw[order]
    starts = np.concatenate([[0], np.cumsum(per_q)[:-1]])
    for q in list(range(0, n, 997)) + [n - 1]:
        want = s[starts[q]: starts[q] + min(per_q[q], k)]
        assert np.array_equal(got[q, : c[q]], want), q
        assert (got[q, c[q]:]["ref_idx"] == 0xFFFFFFFF).all()


def test_topk_empty_and_tiny(hits_10k):
    import hypergen_amd as hg
    ctx = hits_10k[0]
    dev = torch.device("cuda:0")
    out = torch.zeros((5 * 2, 3), dtype=torch.int32, device=dev)
    cnt = torch.full((5,), 7, dtype=torch.int32, device=dev)
    ctx.topk_per_query_dev(0, 0, 5, 2, out.data_ptr(), cnt.data_ptr())
    torch.cuda.synchronize()
    assert int(cnt.sum()) == 0 and bool((out.view(-1, 3)[:, 0] == -1).all())
    h = np.array([(3, 1, 90.0), (2, 1, 95.0), (9, 4, 88.0), (1, 1, 95.0)], hg.ANI_HIT_DTYPE)
    th = torch.from_numpy(h.view(np.int32)).to(dev)
    ctx.topk_per_query_dev(th.data_ptr(), 4, 5, 2, out.data_ptr(), cnt.data_ptr())
    torch.cuda.synchronize()
    got = out.cpu().numpy().view(hg.ANI_HIT_DTYPE).reshape(5, 2)
    assert cnt.tolist() == [0, 2, 0, 0, 1]
    assert [tuple(x) for x in got[1].tolist()] == [(1, 1, 95.0), (2, 1, 95.0)] and got[4][0]["ref_idx"] == 9


@pytest.mark.parametrize("n", [2, 63, 64, 65, 1023, 1024, 4095, 4096, 4097, 8193, 70_001, 300_000])
def test_device_order_on_synthetic_lists(n):
    """the hand-written radix passes against the host order on lists built to stress them: lengths around the wave /
    quarter / tile borders, few distinct ANI values (long runs of ties, i.e. stability), reference indices up to 2^32 - 2
    and a query count whose enumeration key needs more than 32 bits"""
    import hypergen_amd as hg
    rng = np.random.default_rng(n)
    Q = 3_000_000
    h = np.zeros(n, hg.ANI_HIT_DTYPE)
    h["ref_idx"] = rng.integers(0, 2**32 - 2, n, dtype=np.uint64).astype(np.uint32)
    h["ref_idx"][: n // 3] = rng.integers(0, 50, n // 3)
    h["qry_idx"] = rng.integers(0, Q, n)
    h["ani"] = rng.choice(np.array([85.0, 85.00001, 90.5, 99.999, 100.0, 0.0], np.float32), n)
    want = hg.sort_ani_hits(h, Q)
    dev = torch.device("cuda:0")
    with hg.Context(0) as ctx:
        work = torch.from_numpy(h.view(np.int32).copy()).to(dev)
        ctx.sort_ani_hits_dev(work.data_ptr(), n, Q)
        torch.cuda.synchronize()
        got = work.cpu().numpy().view(hg.ANI_HIT_DTYPE).reshape(-1)
        assert np.array_equal(got, want)
        # top-k on the same list (query indices below a small Q so that every query has many hits)
        h2 = h.copy()
        Q2, k = 37, 5
        h2["qry_idx"] = rng.integers(0, Q2, n)
        h2["ref_idx"] = rng.permutation(n).astype(np.uint32)  # distinct, so the expected order is total
        th = torch.from_numpy(h2.view(np.int32).copy()).to(dev)
        out = torch.empty((Q2 * k, 3), dtype=torch.int32, device=dev)
        cnt = torch.empty(Q2, dtype=torch.int32, device=dev)
        ctx.topk_per_query_dev(th.data_ptr(), n, Q2, k, out.data_ptr(), cnt.data_ptr())
        torch.cuda.synchronize()
        got = out.cpu().numpy().view(hg.ANI_HIT_DTYPE).reshape(Q2, k)
        c = cnt.cpu().numpy()
        order = np.lexsort((h2["ref_idx"], -h2["ani"].astype(np.float64), h2["qry_idx"]))
        s = h2[order]
        per_q = np.bincount(h2["qry_idx"], minlength=Q2)
        starts = np.concatenate([[0], np.cumsum(per_q)[:-1]])
        for q in range(Q2):
            m = min(per_q[q], k)
            assert c[q] == m and np.array_equal(got[q, :m], s[starts[q]: starts[q] + m]), q
