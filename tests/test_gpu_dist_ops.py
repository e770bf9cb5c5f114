"""Sharded dist with the reference operands prepared where the rows live (the reference's dist runs on one
host: src/dist.rs:231-294): every "rank" converts ITS reference rows to centred byte operands + control records
(hg_dist_prep_ops_dev), the exchange moves those instead of the i16 rows, hg_dist_block_ops_dev consumes them.  The hit set
must equal the one-call result of hg_dist_dev -- also when the gathered rows are not one contiguous index range (chunked
exchange, d_ref_index), when most rows carry clamped entries, and through the veto -> i16 fallback."""
import numpy as np
import pytest

from conftest import ANI_TOL
import torch

pytestmark = pytest.mark.gpu
D, K = 4096, 21


@pytest.fixture(scope="module")
def hg():
    import hypergen_amd
    return hypergen_amd


def hitset(t, n):
    h = t[: 3 * n].view(-1, 3).cpu().numpy()
    return {(int(a), int(b)): np.int32(c).view(np.float32) for a, b, c in h}


def one_call(hg, hv, n2, th, sym=False):
    n = hv.shape[0]
    cap = 4_000_000
    hits = torch.empty(cap * 3, dtype=torch.int32, device="cuda")
    with hg.Context(0) as ctx:
        found, st = ctx.dist_dev(hv.data_ptr(), n2.data_ptr(), n, hv.data_ptr(), n2.data_ptr(), n, D, K, sym, th, hits.data_ptr(), cap)
        assert st == 0
    return hitset(hits, found)


def prepare_blocks(hg, ctxs, hv, bounds):
    """each rank prepares its rows; returns per-rank (ops, meta, flag) tensors"""
    rb, mb = hg.lib().hg_dist_ops_row_bytes(D), hg.lib().hg_dist_ops_meta_bytes()
    out = []
    for c, (lo, hi) in zip(ctxs, bounds):
        ops = torch.empty((hi - lo, rb), dtype=torch.uint8, device="cuda")
        meta = torch.empty((hi - lo, mb), dtype=torch.uint8, device="cuda")
        flag = torch.full((1,), 77, dtype=torch.int32, device="cuda")
        c.dist_prep_ops_dev(hv[lo:hi].data_ptr(), hi - lo, D, ops.data_ptr(), meta.data_ptr(), flag.data_ptr())
        c.sync()
        out.append((ops, meta, flag))
    return out


def gathered(hg, parts, order_rows=None):
    """what an all-gather leaves on every rank: the blocks back to back (optionally re-ordered row-wise), zero-padded"""
    ops = torch.cat([p[0] for p in parts])
    meta = torch.cat([p[1] for p in parts])
    flags = torch.cat([p[2] for p in parts])
    if order_rows is not None:
        ops, meta = ops[order_rows], meta[order_rows]
    R = ops.shape[0]
    Rp = hg.lib().hg_dist_ops_padded_rows(R)
    buf = torch.full((Rp, ops.shape[1]), 0x5A, dtype=torch.uint8, device="cuda")  # (the call must zero the tail itself)
    buf[:R] = ops
    return buf, meta.contiguous(), flags.contiguous()


@pytest.mark.parametrize("nhash,n,world", [(3333, 3000, 2), (5500, 2600, 3), (3333, 1500, 8)])
def test_prepared_operands_equal_the_one_call_result(hg, nhash, n, world):
    import bench
    from hypergen_amd import shard
    hv = bench.clustered_hvs(n, 0, torch.device("cuda"), n=nhash)
    n2 = (hv.int() ** 2).sum(1).int()
    want = one_call(hg, hv, n2, 85.0)
    bounds = [shard.shard_range(n, r, world) for r in range(world)]
    ctxs = [hg.Context(0) for _ in range(world)]
    try:
        for c in ctxs:
            c.set_debug("dist_path", "i8")  # (small blocks would otherwise not bother with byte operands)
        parts = prepare_blocks(hg, ctxs, hv, bounds)
        assert all(int(p[2][0]) == 0 for p in parts)
        ops, meta, flags = gathered(hg, parts)
        got = {}
        cap = 2_000_000
        hits = torch.empty(cap * 3, dtype=torch.int32, device="cuda")
        for c, (lo, hi) in zip(ctxs, bounds):  # rank r: all references x its own query rows
            found, st = c.dist_block_ops_dev(ops.data_ptr(), meta.data_ptr(), n2.data_ptr(), n, 0, 0, flags.data_ptr(), world,
                                             hv[lo:hi].data_ptr(), n2[lo:hi].data_ptr(), hi - lo, lo, D, K, False, 85.0,
                                             hits.data_ptr(), cap)
            assert st == 0 and c.last_dist_path() == 1
            part = hitset(hits, found)
            assert not (set(part) & set(got))
            got.update(part)
        assert set(got) == set(want)
        assert all(got[k] == want[k] for k in want)  # the same float
        assert bool((ops[n:] == 0).all())  # rows behind R: zeroed by the call
        # more pairs than the hit counter reaches run as blocks of QUERY rows here (the prepared reference block stays whole):
        # the "pair_limit" hook moves that border down
        c, (lo, hi) = ctxs[0], bounds[0]
        c.set_debug("pair_limit", str(n * 97 + 5))
        found, st = c.dist_block_ops_dev(ops.data_ptr(), meta.data_ptr(), n2.data_ptr(), n, 0, 0, flags.data_ptr(), world,
                                         hv[lo:hi].data_ptr(), n2[lo:hi].data_ptr(), hi - lo, lo, D, K, False, 85.0, hits.data_ptr(), cap)
        c.set_debug("pair_limit", "0")
        blocked = hitset(hits, found)
        assert st == 0 and blocked == {k: v for k, v in want.items() if lo <= k[1] < hi}
    finally:
        for c in ctxs:
            c.close()


def test_chunked_exchange_with_a_reference_index_map(hg):
    """two chunks per rank, gathered chunk by chunk: chunk h of every rank back to back -- the gathered rows are not one
    contiguous index range, d_ref_index says which global reference each row is"""
    import bench
    n, world, chunks = 2400, 3, 2
    hv = bench.clustered_hvs(n, 0, torch.device("cuda"), n=3333)
    n2 = (hv.int() ** 2).sum(1).int()
    want = one_call(hg, hv, n2, 86.0)
    rows = n // world
    bounds = [(r * rows, (r + 1) * rows) for r in range(world)]
    ctxs = [hg.Context(0) for _ in range(world)]
    try:
        for c in ctxs:
            c.set_debug("dist_path", "i8")
        parts = prepare_blocks(hg, ctxs, hv, bounds)
        half = rows // chunks
        got = {}
        cap = 1_000_000
        hits = torch.empty(cap * 3, dtype=torch.int32, device="cuda")
        for h in range(chunks):
            idx = torch.cat([torch.arange(r * rows + h * half, r * rows + (h + 1) * half) for r in range(world)]).cuda()
            ops, meta, flags = gathered(hg, parts, order_rows=idx)
            gidx = idx.int().contiguous()
            for c, (lo, hi) in zip(ctxs, bounds):
                found, st = c.dist_block_ops_dev(ops.data_ptr(), meta.data_ptr(), n2[idx].contiguous().data_ptr(), idx.numel(), 0,
                                                 gidx.data_ptr(), flags.data_ptr(), world, hv[lo:hi].data_ptr(), n2[lo:hi].data_ptr(),
                                                 hi - lo, lo, D, K, False, 86.0, hits.data_ptr(), cap)
                assert st == 0
                part = hitset(hits, found)
                assert not (set(part) & set(got))
                got.update(part)
        assert set(got) == set(want) and all(got[k] == want[k] for k in want)
    finally:
        for c in ctxs:
            c.close()


@pytest.mark.parametrize("where", ["owner", "query"])
def test_veto_falls_back_to_the_i16_rows(hg, where):
    """a row that does not fit the byte scheme -- among an owner's reference rows (its flag travels with the gather) or
    among the call's own query rows: HG_ERR_INEXACT, nothing reported; hg_dist_block_dev on the gathered i16 rows gives
    the one-call hits"""
    import bench
    n, world = 2000, 2
    clean = bench.clustered_hvs(n, 0, torch.device("cuda"), n=3333)
    dirty = clean.clone()
    dirty[1500, 5] += 1  # mixed parity in one row: no centred byte form
    ref, qry = (dirty, clean) if where == "owner" else (clean, dirty)
    rn, qn = (ref.int() ** 2).sum(1).int(), (qry.int() ** 2).sum(1).int()
    cap = 1_000_000
    hits = torch.empty(cap * 3, dtype=torch.int32, device="cuda")
    with hg.Context(0) as c0:
        found, st = c0.dist_dev(ref.data_ptr(), rn.data_ptr(), n, qry.data_ptr(), qn.data_ptr(), n, D, K, False, 85.0, hits.data_ptr(), cap)
        assert st == 0
        want = hitset(hits, found)
    bounds = [(0, 1000), (1000, 2000)]
    ctxs = [hg.Context(0) for _ in range(world)]
    try:
        for c in ctxs:
            c.set_debug("dist_path", "i8")
        parts = prepare_blocks(hg, ctxs, ref, bounds)
        ops, meta, flags = gathered(hg, parts)
        fl = flags.cpu().numpy()
        assert (fl != 0).tolist() == ([False, True] if where == "owner" else [False, False])
        got = {}
        for r, (c, (lo, hi)) in enumerate(zip(ctxs, bounds)):
            found, st = c.dist_block_ops_dev(ops.data_ptr(), meta.data_ptr(), rn.data_ptr(), n, 0, 0, flags.data_ptr(), world,
                                             qry[lo:hi].data_ptr(), qn[lo:hi].data_ptr(), hi - lo, lo, D, K, False, 85.0,
                                             hits.data_ptr(), cap)
            if where == "owner" or r == 1:  # every rank sees an owner's flag; only rank 1 holds the dirty query row
                assert st == hg.ERR_INEXACT and found == 0
                found, st = c.dist_block_dev(ref.data_ptr(), rn.data_ptr(), n, 0, qry[lo:hi].data_ptr(), qn[lo:hi].data_ptr(), hi - lo, lo,
                                             D, K, False, 85.0, hits.data_ptr(), cap)
            assert st == 0
            got.update(hitset(hits, found))
        assert set(got) == set(want) and all(got[k] == want[k] for k in want)
    finally:
        for c in ctxs:
            c.close()


@pytest.mark.parametrize("seed", range(6))
def test_prepared_operands_random_ragged_shapes(hg, orc, seed):
    """small and ragged: a handful of rows, hv_d that is not a multiple of 128, one or several owners, planted clamped
    entries -- against the oracle's ANI matrix"""
    rng = np.random.default_rng(500 + seed)
    d = int(rng.choice([64, 200, 1000, 1032, 4096]))
    R, Q, world = int(rng.integers(1, 400)), int(rng.integers(1, 300)), int(rng.integers(1, 4))
    n = int(rng.choice([61, 600, 3333]))
    cnt = rng.binomial(n, 0.5, (R + Q, d))
    hv = (2 * cnt - n).astype(np.int16)
    for _ in range(int(rng.integers(0, 12))):  # clamped entries: |count - n / 2| > 127
        hv[rng.integers(0, R + Q), rng.integers(0, d)] += np.int16(2 * int(rng.choice([-1, 1])) * int(rng.integers(130, 200)))
    ref, qry = hv[:R], hv[R:]
    rn = np.array([orc.hv_norm2(x) for x in ref], np.int32)
    qn = np.array([orc.hv_norm2(x) for x in qry], np.int32)
    want = orc.ani_matrix(ref, rn, qry, qn, 21)
    th = float(rng.choice([0.0, 50.0, 70.0]))
    t_ref, t_rn = torch.from_numpy(ref).cuda(), torch.from_numpy(rn).cuda()
    t_qry, t_qn = torch.from_numpy(qry).cuda(), torch.from_numpy(qn).cuda()
    from hypergen_amd import shard
    bounds = [shard.shard_range(R, r, world) for r in range(world)]
    rb, mb = hg.lib().hg_dist_ops_row_bytes(d), hg.lib().hg_dist_ops_meta_bytes()
    with hg.Context(0) as c:
        parts = []
        for lo, hi in bounds:
            ops = torch.empty((max(hi - lo, 1), rb), dtype=torch.uint8, device="cuda")
            meta = torch.empty((max(hi - lo, 1), mb), dtype=torch.uint8, device="cuda")
            flag = torch.zeros(1, dtype=torch.int32, device="cuda")
            c.dist_prep_ops_dev(t_ref[lo:hi].contiguous().data_ptr() if hi > lo else 0, hi - lo, d, ops.data_ptr(), meta.data_ptr(), flag.data_ptr())
            parts.append((ops[: hi - lo], meta[: hi - lo], flag))
        ops, meta, flags = gathered(hg, parts)
        cap = R * Q + 16
        hits = torch.empty(cap * 3, dtype=torch.int32, device="cuda")
        found, st = c.dist_block_ops_dev(ops.data_ptr(), meta.data_ptr(), t_rn.data_ptr(), R, 0, 0, flags.data_ptr(), world,
                                         t_qry.data_ptr(), t_qn.data_ptr(), Q, 0, d, 21, False, th, hits.data_ptr(), cap)
        if st == hg.ERR_INEXACT:  # a residual beyond a byte / too many entries in one row: the i16 path must agree with the oracle
            found, st = c.dist_block_dev(t_ref.data_ptr(), t_rn.data_ptr(), R, 0, t_qry.data_ptr(), t_qn.data_ptr(), Q, 0, d, 21, False, th,
                                         hits.data_ptr(), cap)
        assert st == 0
        got = hitset(hits, found)
    sure = {(i, j) for i, j in zip(*np.nonzero(want >= th + ANI_TOL))}
    maybe = {(i, j) for i, j in zip(*np.nonzero(want >= th - ANI_TOL))}
    assert sure <= set(got) <= maybe
    assert all(abs(float(v) - float(want[k])) <= ANI_TOL for k, v in got.items())
