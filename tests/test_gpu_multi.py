"""The sharded (multi-GPU) paths through the REAL library on a one-GPU box: two hg_ctx objects on device 0 act as
two ranks (process-per-GPU pattern of bench.py / shard.py), and hg_multi with device_ids = {0, 0} runs the
in-process pattern (peer all-gather, merged hits).  Everything must equal the one-context results bit for bit."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

D = 4096


def _key(h):
    return np.sort(h, order=["ref_idx", "qry_idx"])


@pytest.fixture(scope="module")
def hg():
    import hypergen_amd
    return hypergen_amd


@pytest.fixture(scope="module")
def genomes(orc):
    # 10 genomes of two clusters (members 0..4 of clusters 0 and 1), ragged lengths
    return [orc.synth_genome(g, 150_000 + 7_001 * i) for i, g in enumerate((0, 1, 2, 3, 4, 100, 101, 102, 103, 104))]


def test_two_contexts_as_two_ranks_sketch_then_dist_blocks(hg, orc, genomes):
    """bench.py's decomposition with the real kernels: rank r sketches shard_range(n, r, 2) on its own ctx, the
    reference matrix is the concatenation (what the all-gather assembles), each rank computes the block
    (all refs) x (its query rows) with global indices."""
    from hypergen_amd import shard
    p = hg.default_params(scaled=100)
    n, world = len(genomes), 2
    with hg.Context(0) as one:
        hv1, n21, nh1 = one.sketch_batch(genomes, p)
        want = one.dist(hv1, n21, hv1, n21, 21, symmetric=False, ani_th=80.0)
        want_sym = one.dist(hv1, n21, hv1, n21, 21, symmetric=True, ani_th=80.0)
    ranks = [hg.Context(0) for _ in range(world)]
    try:
        parts = []
        for r, ctx in enumerate(ranks):
            lo, hi = shard.shard_range(n, r, world)
            parts.append(ctx.sketch_batch(genomes[lo:hi], p))
        hv = np.concatenate([x[0] for x in parts])
        n2 = np.concatenate([x[1] for x in parts])
        nh = np.concatenate([x[2] for x in parts])
        assert np.array_equal(hv, hv1) and np.array_equal(n2, n21) and np.array_equal(nh, nh1)
        for g in (0, 7):
            w_hv, w_n2, w_nh = orc.sketch_genome(genomes[g], scaled=100)
            assert nh[g] == w_nh and n2[g] == w_n2 and np.array_equal(hv[g], w_hv)
        dev = torch.device("cuda:0")
        t_hv, t_n2 = torch.from_numpy(hv).to(dev), torch.from_numpy(n2).to(dev)
        for sym, ref in ((False, want), (True, want_sym)):
            got = []
            for r, ctx in enumerate(ranks):
                lo, hi = shard.shard_range(n, r, world)
                out = torch.zeros(3 * 4096, dtype=torch.int32, device=dev)
                q_hv, q_n2 = t_hv[lo:hi].contiguous(), t_n2[lo:hi].contiguous()
                found, st = ctx.dist_block_dev(t_hv.data_ptr(), t_n2.data_ptr(), n, 0, q_hv.data_ptr(), q_n2.data_ptr(),
                                               hi - lo, lo, D, 21, sym, 80.0, out.data_ptr(), 4096)
                assert st == 0
                got.append(out[: 3 * found].cpu().numpy().view(hg.ANI_HIT_DTYPE))
            got = np.concatenate(got)
            assert got.size == ref.size and got.size > 10
            assert np.array_equal(_key(got), _key(ref))  # same pairs, same float bits
    finally:
        for c in ranks:
            c.close()


@pytest.mark.parametrize("ids", [(0,), (0, 0), (0, 0, 0)])
def test_hg_multi_equals_single_context(hg, orc, genomes, ids):
    p = hg.default_params(scaled=100)
    with hg.Context(0) as one:
        hv1, n21, nh1 = one.sketch_batch(genomes, p)
        with hg.Multi(list(ids)) as m:
            hv, n2, nh = m.sketch_batch(genomes, p)
            assert np.array_equal(hv, hv1) and np.array_equal(n2, n21) and np.array_equal(nh, nh1)
            for sym in (False, True):
                want = one.dist(hv1, n21, hv1, n21, 21, symmetric=sym, ani_th=80.0)
                got = m.dist(hv1, n21, None, None, 21, symmetric=sym, ani_th=80.0, cap=7)  # cap grows on demand
                assert got.size == want.size and np.array_equal(_key(got), _key(want)), (ids, sym)
            # distinct reference and query sets, ragged shard sizes
            r_hv, r_n2, q_hv, q_n2 = hv1[:7], n21[:7], hv1[3:], n21[3:]
            want = one.dist(r_hv, r_n2, q_hv, q_n2, 21, symmetric=False, ani_th=80.0)
            got = m.dist(r_hv, r_n2, q_hv, q_n2, 21, symmetric=False, ani_th=80.0)
            assert np.array_equal(_key(got), _key(want))


def test_hg_multi_dev_resident_shards_larger(hg):
    """Resident shards (the sketch -> dist flow without a host round trip): 3 000 clustered HVs held as two shards
    on 'two devices', all-gathered by peer copies, symmetric and not; and a separate query set."""
    import bench
    dev = torch.device("cuda:0")
    n = 3000
    hv = bench.clustered_hvs(n, 0, dev)
    n2 = (hv.int() ** 2).sum(1).int()
    with hg.Context(0) as one, hg.Multi([0, 0]) as m:
        cut = 1700  # uneven shards
        parts = [(hv[:cut].contiguous(), n2[:cut].contiguous()), (hv[cut:].contiguous(), n2[cut:].contiguous())]
        rows = [cut, n - cut]
        for sym in (False, True):
            out = torch.zeros(3 * 400_000, dtype=torch.int32, device=dev)
            found, st = one.dist_dev(hv.data_ptr(), n2.data_ptr(), n, hv.data_ptr(), n2.data_ptr(), n, D, 21, sym, 85.0,
                                     out.data_ptr(), 400_000)
            torch.cuda.synchronize()
            want = out[: 3 * found].cpu().numpy().view(hg.ANI_HIT_DTYPE)
            got = m.dist_dev([x[0].data_ptr() for x in parts], [x[1].data_ptr() for x in parts], rows, None, None, None,
                             D, 21, sym, 85.0, cap=400_000)
            assert found > 10_000 and got.size == found and np.array_equal(_key(got), _key(want)), sym
        # separate query shards
        q = bench.clustered_hvs(900, 100, dev)
        qn = (q.int() ** 2).sum(1).int()
        out = torch.zeros(3 * 400_000, dtype=torch.int32, device=dev)
        found, st = one.dist_dev(hv.data_ptr(), n2.data_ptr(), n, q.data_ptr(), qn.data_ptr(), 900, D, 21, False, 85.0,
                                 out.data_ptr(), 400_000)
        want = out[: 3 * found].cpu().numpy().view(hg.ANI_HIT_DTYPE)
        qparts = [(q[:300].contiguous(), qn[:300].contiguous()), (q[300:].contiguous(), qn[300:].contiguous())]
        got = m.dist_dev([x[0].data_ptr() for x in parts], [x[1].data_ptr() for x in parts], rows,
                         [x[0].data_ptr() for x in qparts], [x[1].data_ptr() for x in qparts], [300, 600], D, 21, False,
                         85.0, cap=400_000)
        assert found > 1000 and np.array_equal(_key(got), _key(want))


def test_hg_multi_exchanges_prepared_operands_and_falls_back_to_i16_rows(hg):
    """the exchange step moves byte operands + control records prepared by the shard that owns the rows (half the bytes of
    the i16 rows); rows that do not fit the byte scheme -- on an owner's or on a query side -- bring the i16 exchange, and
    so does the "f16" hook on shard 0: identical hits every way"""
    import bench
    dev = torch.device("cuda:0")
    n = 2400
    hv = bench.clustered_hvs(n, 0, dev).clone()
    for case in ("clean", "dirty", "hook"):
        if case == "dirty":
            hv[2000, 9] += 1  # mixed parity: no centred byte form for this row
        n2 = (hv.int() ** 2).sum(1).int()
        with hg.Context(0) as one, hg.Multi([0, 0, 0]) as m:
            if case == "hook":
                hg.lib().hg_ctx_set_debug(m.ctx_handle(0), b"dist_path", b"f16")
            rows = [900, 700, 800]
            cuts = [0, 900, 1600, 2400]
            parts = [(hv[a:b].contiguous(), n2[a:b].contiguous()) for a, b in zip(cuts[:-1], cuts[1:])]
            for sym in (False, True):
                out = torch.zeros(3 * 300_000, dtype=torch.int32, device=dev)
                found, st = one.dist_dev(hv.data_ptr(), n2.data_ptr(), n, hv.data_ptr(), n2.data_ptr(), n, D, 21, sym, 85.0,
                                         out.data_ptr(), 300_000)
                want = out[: 3 * found].cpu().numpy().view(hg.ANI_HIT_DTYPE)
                got = m.dist_dev([x[0].data_ptr() for x in parts], [x[1].data_ptr() for x in parts], rows, None, None, None,
                                 D, 21, sym, 85.0, cap=300_000)
                assert found > 5_000 and np.array_equal(_key(got), _key(want)), (case, sym)
                rep = m.gather_report()
                assert ("prepared byte operands" in rep) == (case == "clean"), (case, rep)


def test_hg_multi_hamming_search_sharded_refs(hg, orc):
    rng = np.random.default_rng(77)
    HD, R, Q = 2048, 1501, 64
    refs = rng.integers(0, 2**32, (R, HD // 32), dtype=np.uint64).astype(np.uint32)
    q = refs[rng.integers(0, R, Q)].copy()
    q[:, ::3] ^= np.uint32(0x00010001)
    d = orc.hamming_matrix(refs, q)
    ri, qi = np.nonzero(d <= 60)
    want = sorted(zip(ri.tolist(), qi.tolist(), d[ri, qi].tolist()))
    for ids in ((0,), (0, 0), (0, 0, 0, 0)):
        with hg.Multi(list(ids)) as m:
            h = m.hamming_search(refs, q, HD, 60, cap=16)
            assert sorted(zip(h["ref_idx"].tolist(), h["qry_idx"].tolist(), h["dist"].tolist())) == want, ids


def test_hg_multi_rccl_gather_one_rank_and_reports(hg, genomes):
    """The exchange step of dist through RCCL behind the C ABI (hg_multi_set_gather(HG_GATHER_RCCL)): on this one-GPU box a
    one-rank communicator (ncclCommInitAll + ncclAllGather / grouped ncclBroadcast really execute); repeated device ids
    are refused (one communicator rank per GPU); both gathers give the same hits."""
    p = hg.default_params(scaled=100)
    with hg.Context(0) as one:
        hv1, n21, _ = one.sketch_batch(genomes, p)
        want = one.dist(hv1, n21, hv1, n21, 21, symmetric=True, ani_th=80.0)
        want2 = one.dist(hv1[:7], n21[:7], hv1[3:], n21[3:], 21, symmetric=False, ani_th=80.0)
    with hg.Multi([0]) as m:
        assert m.gather_mode() == hg.GATHER_PEER and "0 of 0 ordered device pairs" in m.peer_report()
        got = m.dist(hv1, n21, None, None, 21, symmetric=True, ani_th=80.0)
        assert "peer pulls" in m.gather_report()
        m.set_gather(hg.GATHER_RCCL)
        assert m.gather_mode() == hg.GATHER_RCCL
        got_r = m.dist(hv1, n21, None, None, 21, symmetric=True, ani_th=80.0)
        assert m.gather_report().startswith("rccl ncclAllGather over 1 ranks"), m.gather_report()
        got_r2 = m.dist(hv1[:7], n21[:7], hv1[3:], n21[3:], 21, symmetric=False, ani_th=80.0)
        assert np.array_equal(_key(got), _key(want)) and np.array_equal(_key(got_r), _key(want))
        assert np.array_equal(_key(got_r2), _key(want2))
        m.set_gather(hg.GATHER_PEER)
        assert np.array_equal(_key(m.dist(hv1, n21, None, None, 21, symmetric=True, ani_th=80.0)), _key(want))
    with hg.Multi([0, 0]) as m2:
        with pytest.raises(hg.HgError) as e:
            m2.set_gather(hg.GATHER_RCCL)
        assert e.value.status == hg.ERR_UNSUPPORTED and m2.gather_mode() == hg.GATHER_PEER
        assert "0 of 0" in m2.peer_report()  # the two shards share one device: no peer pair to enable
