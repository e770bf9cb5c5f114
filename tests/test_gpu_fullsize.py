"""BASELINE-size runs checked through size-independent properties (the oracle cannot redo 1 000 x 5 Mbp
in seconds): determinism, sampled full parity, norm / dot identities, ANI symmetry, hit-set consistency."""
import os

import numpy as np
import pytest

from conftest import ANI_TOL
import torch

pytestmark = pytest.mark.gpu

N, L, D = 1000, 5_000_000, 4096


@pytest.fixture(scope="module")
def sketched(orc):
    import hypergen_amd as hg
    dev = torch.device("cuda:0")
    ctx = hg.Context(0)
    ctx.set_stream(torch.cuda.current_stream().cuda_stream)
    stride = (L + 1 + 15) // 16 * 16
    seq = torch.empty(N * stride + 64, dtype=torch.uint8, device=dev)
    ctx.synth_genomes_dev(0, N, L, stride, seq.data_ptr())
    offs = np.arange(N, dtype=np.uint64) * stride
    lens = np.full(N, L + 1, np.uint64)
    p = hg.default_params()
    out = []
    for _ in range(2):
        hv = torch.empty((N, D), dtype=torch.int16, device=dev)
        n2 = torch.empty(N, dtype=torch.int32, device=dev)
        nh = torch.empty(N, dtype=torch.int32, device=dev)
        ctx.sketch_batch_dev(seq.data_ptr(), offs, lens, p, hv.data_ptr(), n2.data_ptr(), nh.data_ptr())
        ctx.sync()  # (reads the step's check word, then waits for the stream)
        out.append((hv, n2, nh))
    yield ctx, seq, stride, out
    ctx.close()


def test_sketch_1000x5mbp_properties(sketched, orc):
    ctx, seq, stride, out = sketched
    (hv, n2, nh), (hv_b, n2_b, nh_b) = out
    # idempotence: two runs are bit-identical although hits are appended in nondeterministic order
    assert torch.equal(hv, hv_b) and torch.equal(n2, n2_b) and torch.equal(nh, nh_b)
    nhc = nh.cpu().numpy()
    assert 3100 < nhc.min() and nhc.max() < 3600 and abs(nhc.mean() - (L - 20) / 1500) < 60  # 10 independent cluster roots
    # norm identity (i32) and the bundling identity sum_d hv[d] == 2*ones - n*D with parity n (mod 2)
    assert torch.equal((hv.int() ** 2).sum(1).int(), n2)
    assert bool(((hv.int() - nh[:, None]) % 2 == 0).all())  # hv[d] = 2*count - n
    # device generator == host generator, and full parity for a sample of genomes spread over the batch
    for g in (0, 1, 137, 500, 999):
        host = orc.synth_genome(g, L)
        assert np.array_equal(seq[g * stride: g * stride + L + 1].cpu().numpy(), host)
        w_hv, w_n2, w_nh = orc.sketch_genome(host)
        assert nhc[g] == w_nh and int(n2[g]) == w_n2 and np.array_equal(hv[g].cpu().numpy(), w_hv), g


def test_dist_of_the_1000_sketches_properties(sketched, orc):
    ctx, seq, stride, out = sketched
    hv, n2, nh = out[0]
    dev = hv.device
    full = torch.empty((N, N), dtype=torch.float32, device=dev)
    ctx.dist_full_dev(hv.data_ptr(), n2.data_ptr(), N, hv.data_ptr(), n2.data_ptr(), N, D, 21, full.data_ptr())
    torch.cuda.synchronize()
    assert torch.equal(full, full.T)                       # exact integer dots -> exactly symmetric
    assert bool((full.diagonal() == 100.0).all())
    # exact dots: recompute a block with int64 matmul on the device and the reference's float32 formula
    blk = slice(100, 228)
    dots = (hv[blk].double() @ hv.double().T).round().long()  # exact: |dot| << 2^53
    den = (n2[blk].long()[:, None] + n2.long()[None, :] - dots).float()
    j = dots.float() / den
    ani = (1.0 + torch.log(2.0 / (1.0 / j + 1.0)) / 21.0)
    ani = torch.nan_to_num(ani, nan=0.0).clamp(0.0, 1.0) * 100.0
    assert float((full[blk] - ani).abs().max()) <= 1e-4
    # clustered workload: members of a cluster (100 consecutive genomes, <= 9.9 % substitutions) are
    # close, different clusters are far
    # (member m differs from its root by 0.1*m %, two members by up to ~20 %)
    # unrelated genomes sit at the estimator's noise floor (chance dot > 0 => up to ~87 % at D=4096, n~3333)
    assert float(full[0, :100].min()) > 89.0 and float(full[:100, 100:].max()) < 90.0
    # thresholded path == thresholded full matrix
    cap = 1 << 20
    hits = torch.empty(cap * 3, dtype=torch.int32, device=dev)
    n, st = ctx.dist_dev(hv.data_ptr(), n2.data_ptr(), N, hv.data_ptr(), n2.data_ptr(), N, D, 21, True, 95.0,
                         hits.data_ptr(), cap)
    assert st == 0
    iu = torch.triu(torch.ones((N, N), dtype=torch.bool, device=dev), 1)
    assert n == int(((full >= 95.0) & iu).sum()) and n > 1000  # the same float in both kernels: the same set
    h = hits[: 3 * n].view(-1, 3)
    vals = h[:, 2].view(torch.float32)
    assert float((full[h[:, 0].long(), h[:, 1].long()] - vals).abs().max()) == 0.0  # same kernel arithmetic
    # model check of the estimator itself: substitution rate p -> ANI ~ 100*(1-p)
    est = full[0, 1:100].cpu().numpy()
    true = 100.0 * (1.0 - 0.001 * np.arange(1, 100))
    assert np.abs(est - true).max() < 1.0


def test_host_fed_batch_in_several_uploads_equals_resident(sketched, orc):
    """hg_sketch_batch cuts a host batch into ~64 MiB sub-batches whose uploads overlap the kernels; the
    result must equal the device-resident run of the same genomes, for pageable and for pinned sources and
    for ragged lengths (the last genome of a sub-batch differs from the first of the next)."""
    ctx, seq, stride, out = sketched
    hv, n2, nh = out[0]
    import hypergen_amd as hg
    p = hg.default_params()
    pick = list(range(0, 48))  # 48 x 5 Mbp = 240 MB: four sub-batches
    host = seq.cpu().numpy()
    rows = [np.ascontiguousarray(host[g * stride: g * stride + L + 1]) for g in pick]
    for label in ("pageable", "pinned"):
        src = rows if label == "pageable" else [torch.from_numpy(r).pin_memory().numpy() for r in rows]
        h_hv, h_n2, h_nh = ctx.sketch_batch(src, p)
        assert np.array_equal(h_hv, hv[pick].cpu().numpy()), label
        assert np.array_equal(h_n2, n2[pick].cpu().numpy()) and np.array_equal(h_nh, nh[pick].cpu().numpy()), label
    # ragged: truncated copies, checked against the oracle on a sample
    ragged = [rows[i][: 1 + (i * 977_773) % L] for i in range(len(rows))]
    r_hv, r_n2, r_nh = ctx.sketch_batch(ragged, p)
    for i in (0, 1, 13, 14, 15, 47):
        w_hv, w_n2, w_nh = orc.sketch_genome(ragged[i])
        assert r_nh[i] == w_nh and r_n2[i] == w_n2 and np.array_equal(r_hv[i], w_hv), i


def _independent_block(orc, hv, n2, rows, k=21):
    """ANI of rows `rows` against every column from two sources that share no code with the kernels: the CPU oracle
    (orc_ani_matrix: scalar i16 dot per pair, src/dist.rs:139-161) and an fp64 torch GEMM (exact dots: |dot| << 2^53)
    pushed through the reference's float32 formula (src/dist.rs:153-160).  Returns both as float32 tensors on the device."""
    hvc, n2c = hv.cpu().numpy(), n2.cpu().numpy()
    o = torch.from_numpy(orc.ani_matrix(hvc[rows], n2c[rows], hvc, n2c, k)).to(hv.device)
    dots = (hv[rows].double() @ hv.double().T).round().long()
    den = (n2[rows].long()[:, None] + n2.long()[None, :] - dots).int().float()  # i32 wrapping sum, then f32
    j = dots.float() / den
    a = 1.0 + torch.log(2.0 / (1.0 / j + 1.0)) / float(k)
    a = torch.nan_to_num(a, nan=0.0).clamp(0.0, 1.0) * 100.0
    return o, a


def _check_hits_against_block(h, rows, indep, th, sym, tol):
    """thresholded hits whose reference row lies in `rows` == the independent matrix thresholded, reported ANI == the independent
    value -- with tol = 0 for the CPU oracle; the fp64 / torch.log model is given 1e-4 (pairs that close to the threshold may
    fall either way)"""
    lo, hi = rows.start, rows.stop
    sel = (h[:, 0] >= lo) & (h[:, 0] < hi)
    ri, qi, ani = h[sel, 0].long() - lo, h[sel, 1].long(), h[sel, 2].contiguous().view(torch.float32)
    got = torch.zeros_like(indep, dtype=torch.bool)
    got[ri, qi] = True
    live = torch.ones_like(got)
    if sym:
        live = torch.arange(lo, hi, device=indep.device)[:, None] < torch.arange(indep.shape[1], device=indep.device)[None, :]
    sure_in, sure_out = (indep >= th + tol) & live, (indep < th - tol) | ~live
    assert bool(got[sure_in].all()) and not bool(got[sure_out].any())
    assert float((ani - indep[ri, qi]).abs().max()) <= tol
    return int(sel.sum())


@pytest.mark.parametrize("tile,nhash", [("", 3333), ("big", 3333), ("wide", 3333), ("nt3", 3333), ("small", 3333),
                                        ("", 6666), ("small", 6666), ("", 20000)])
def test_dist_10k_thresholded_equals_full_matrix(orc, tile, nhash):
    """BASELINE configs[3] size: the thresholded entry point (speculative schedule, 256-wide / 320-wide / 192-wide
    / 128 tiles, LDS-DMA with loader waves, phase-0 filter, one reservation per workgroup) must report exactly the
    pairs whose ANI in the full-matrix mode (a different kernel variant) reaches the threshold, with the same
    float, for an asymmetric and a symmetric call.  Kernel-vs-kernel is not the only check: a 512-row block that
    straddles the diagonal and a cluster boundary is compared -- full matrix AND the hits of every tile variant -- with
    the CPU oracle and with an fp64 torch GEMM through the reference's float32 formula.  nhash > 4096: sketches whose dot products need several exact
    f32 accumulation windows (speculation vetoed, windowed statistics, i32 side accumulators, 256 x 192 tiles)."""
    import os
    import bench
    import hypergen_amd as hg
    dev = torch.device("cuda:0")
    ctx = hg.Context(0)
    ctx.set_stream(torch.cuda.current_stream().cuda_stream)
    n = 10000
    hv = bench.clustered_hvs(n, 0, dev, n=nhash)
    n2 = (hv.int() ** 2).sum(1).int()
    full = torch.empty((n, n), dtype=torch.float32, device=dev)
    ctx.dist_full_dev(hv.data_ptr(), n2.data_ptr(), n, hv.data_ptr(), n2.data_ptr(), n, D, 21, full.data_ptr())
    torch.cuda.synchronize()
    rows = slice(3900, 4412)
    ind_orc, ind_f64 = _independent_block(orc, hv, n2, rows)
    assert float((full[rows] - ind_orc).abs().max()) <= ANI_TOL and float((full[rows] - ind_f64).abs().max()) <= 1e-4
    cap = 4_000_000
    hits = torch.empty(cap * 3, dtype=torch.int32, device=dev)
    try:
        ctx.set_debug("dist_tile", tile)
        for sym, th in ((False, 85.0), (True, 86.5)):
            found, st = ctx.dist_dev(hv.data_ptr(), n2.data_ptr(), n, hv.data_ptr(), n2.data_ptr(), n, D, 21, sym, th,
                                     hits.data_ptr(), cap)
            torch.cuda.synchronize()
            assert st == 0 and 0 < found < cap
            h = hits[: found * 3].view(found, 3)
            ri, qi, ani = h[:, 0].long(), h[:, 1].long(), h[:, 2].view(torch.float32)
            want = full >= th
            if sym:
                want = torch.triu(want, diagonal=1)
            assert found == int(want.sum())
            got = torch.zeros_like(want)
            got[ri, qi] = True
            assert bool((got == want).all())
            assert float((ani - full[ri, qi]).abs().max()) <= ANI_TOL
            for indep, tol in ((ind_orc, ANI_TOL), (ind_f64, 1e-4)):  # ... and not only against another variant of the same kernel
                assert _check_hits_against_block(h, rows, indep, th, sym, tol) > 0
    finally:
        ctx.close()


def test_dist_windowed_192_wide_tiles_ragged_q():
    """Q = 4993 needs 5 184 zero-padded query rows for the 192-wide tiles of the windowed big geometry (more than
    the 256- and 320-row paddings give): the LDS-DMA must read zeros there, not stale memory -- first with a
    workspace that a larger earlier call sized and filled, then compared with the full-matrix mode."""
    import bench
    import hypergen_amd as hg
    dev = torch.device("cuda:0")
    ctx = hg.Context(0)
    ctx.set_stream(torch.cuda.current_stream().cuda_stream)
    try:
        big = bench.clustered_hvs(5600, 0, dev, n=6666)
        bn2 = (big.int() ** 2).sum(1).int()
        cap = 4_000_000
        hits = torch.empty(cap * 3, dtype=torch.int32, device=dev)
        # 1) a larger call fills the f16 workspaces with non-zero rows up to 5 600
        ctx.dist_dev(big.data_ptr(), bn2.data_ptr(), 5600, big.data_ptr(), bn2.data_ptr(), 5600, D, 21, False, 85.0,
                     hits.data_ptr(), cap)
        R, Q = 3400, 4993
        r, q = big[:R].contiguous(), big[200:200 + Q].contiguous()
        rn, qn = bn2[:R].contiguous(), bn2[200:200 + Q].contiguous()
        full = torch.empty((R, Q), dtype=torch.float32, device=dev)
        ctx.dist_full_dev(r.data_ptr(), rn.data_ptr(), R, q.data_ptr(), qn.data_ptr(), Q, D, 21, full.data_ptr())
        torch.cuda.synchronize()
        for tile in ("big", ""):
            ctx.set_debug("dist_tile", tile)
            found, st = ctx.dist_dev(r.data_ptr(), rn.data_ptr(), R, q.data_ptr(), qn.data_ptr(), Q, D, 21, False, 85.0,
                                     hits.data_ptr(), cap)
            torch.cuda.synchronize()
            assert st == 0 and 0 < found < cap
            h = hits[: found * 3].view(found, 3)
            ri, qi, ani = h[:, 0].long(), h[:, 1].long(), h[:, 2].view(torch.float32)
            want = full >= 85.0
            assert found == int(want.sum())
            got = torch.zeros_like(want)
            got[ri, qi] = True
            assert bool((got == want).all())
            assert float((ani - full[ri, qi]).abs().max()) <= ANI_TOL
    finally:
        ctx.close()


def test_work_is_ordered_with_the_callers_stream():
    """Inputs produced by the caller's kernels on a stream (default and non-default) and consumed by the ctx
    right away, no synchronisation in between: results must equal the synchronised run."""
    import hypergen_amd as hg
    dev = torch.device("cuda:0")
    n = 6000
    ctx = hg.Context(0)
    try:
        want = None
        for use_side_stream in (False, True, False):
            side = torch.cuda.Stream() if use_side_stream else torch.cuda.current_stream()
            with torch.cuda.stream(side):
                ctx.set_stream(side.cuda_stream)
                g = torch.Generator(device=dev)
                g.manual_seed(11)
                base = torch.randint(-60, 60, (60, D), generator=g, device=dev, dtype=torch.int32)
                hv = (base[torch.arange(n, device=dev) % 60] + torch.randint(-25, 25, (n, D), generator=g, device=dev,
                                                                            dtype=torch.int32)).to(torch.int16)
                n2 = (hv.int() ** 2).sum(1).int()
                hits = torch.empty(3 * 2_000_000, dtype=torch.int32, device=dev)
                found, st = ctx.dist_dev(hv.data_ptr(), n2.data_ptr(), n, hv.data_ptr(), n2.data_ptr(), n, D, 21, True, 90.0,
                                         hits.data_ptr(), 2_000_000)
                side.synchronize()
            assert st == 0 and found > 1000
            if want is None:
                torch.cuda.synchronize()
                found2, _ = ctx.dist_dev(hv.data_ptr(), n2.data_ptr(), n, hv.data_ptr(), n2.data_ptr(), n, D, 21, True, 90.0,
                                         hits.data_ptr(), 2_000_000)
                want = found2
            assert found == want, (use_side_stream, found, want)
    finally:
        ctx.close()


def test_config2_sketch_10k_x_5mbp_resident_and_sharded(orc):
    """BASELINE configs[2]: 10 000 synthetic 5 Mbp genomes (50 GB resident on one MI355X).  Checked through
    size-independent properties + sampled oracle parity, then as the 8-way sharded job: the shard a rank would own
    (shard_range(10 000, r, 8)), sketched on its own from its own generator call, must equal the same rows of the
    one-GPU run bit for bit; finally the 10 000 real sketches go through dist (configs[3] on real data)."""
    import hypergen_amd as hg
    from hypergen_amd import shard
    dev = torch.device("cuda:0")
    n10 = 10_000
    ctx = hg.Context(0)
    ctx.set_stream(torch.cuda.current_stream().cuda_stream)
    try:
        stride = (L + 1 + 15) // 16 * 16
        seq = torch.empty(n10 * stride + 64, dtype=torch.uint8, device=dev)
        ctx.synth_genomes_dev(0, n10, L, stride, seq.data_ptr())
        offs = np.arange(n10, dtype=np.uint64) * stride
        lens = np.full(n10, L + 1, np.uint64)
        p = hg.default_params()
        hv = torch.empty((n10, D), dtype=torch.int16, device=dev)
        n2 = torch.empty(n10, dtype=torch.int32, device=dev)
        nh = torch.empty(n10, dtype=torch.int32, device=dev)
        ctx.sketch_batch_dev(seq.data_ptr(), offs, lens, p, hv.data_ptr(), n2.data_ptr(), nh.data_ptr())
        ctx.sync()  # (reads the step's check word, then waits for the stream)
        nhc = nh.cpu().numpy()
        assert 3050 < nhc.min() and nhc.max() < 3650 and abs(nhc.mean() - (L - 20) / 1500) < 30
        assert torch.equal((hv.int() ** 2).sum(1).int(), n2)
        assert bool(((hv.int() - nh[:, None]) % 2 == 0).all())
        for g in (0, 2_345, 5_000, 7_777, 9_999):  # sampled full parity with the CPU oracle
            host = orc.synth_genome(g, L)
            assert np.array_equal(seq[g * stride: g * stride + L + 1].cpu().numpy(), host)
            w_hv, w_n2, w_nh = orc.sketch_genome(host)
            assert nhc[g] == w_nh and int(n2[g]) == w_n2 and np.array_equal(hv[g].cpu().numpy(), w_hv), g
        # the sharded job, ranks 0, 3 and 7 of 8 (each from scratch: own generator call, own batch plan)
        for r in (0, 3, 7):
            lo, hi = shard.shard_range(n10, r, 8)
            m = hi - lo
            s_seq = torch.empty(m * stride + 64, dtype=torch.uint8, device=dev)
            ctx.synth_genomes_dev(lo, m, L, stride, s_seq.data_ptr())
            s_hv = torch.empty((m, D), dtype=torch.int16, device=dev)
            s_n2 = torch.empty(m, dtype=torch.int32, device=dev)
            s_nh = torch.empty(m, dtype=torch.int32, device=dev)
            ctx.sketch_batch_dev(s_seq.data_ptr(), offs[:m], lens[:m], p, s_hv.data_ptr(), s_n2.data_ptr(), s_nh.data_ptr())
            ctx.sync()  # (reads the step's check word, then waits for the stream)
            assert torch.equal(s_hv, hv[lo:hi]) and torch.equal(s_n2, n2[lo:hi]) and torch.equal(s_nh, nh[lo:hi]), r
            del s_seq, s_hv
        del seq
        # dist of the 10 000 real sketches: thresholded output == thresholded full matrix on a 2 048-row block
        cap = 4_000_000
        hits = torch.empty(cap * 3, dtype=torch.int32, device=dev)
        found, st = ctx.dist_dev(hv.data_ptr(), n2.data_ptr(), n10, hv.data_ptr(), n2.data_ptr(), n10, D, 21, True, 90.0,
                                 hits.data_ptr(), cap)
        torch.cuda.synchronize()
        assert st == 0 and 100_000 < found < cap  # 100 clusters x ~C(100,2) close pairs
        blk = 2048
        full = torch.empty((blk, n10), dtype=torch.float32, device=dev)
        ctx.dist_full_dev(hv.data_ptr(), n2.data_ptr(), blk, hv.data_ptr(), n2.data_ptr(), n10, D, 21, full.data_ptr())
        torch.cuda.synchronize()
        h = hits[: found * 3].view(found, 3)
        sel = h[:, 0] < blk
        ri, qi, ani = h[sel, 0].long(), h[sel, 1].long(), h[sel, 2].contiguous().view(torch.float32)
        want = torch.triu(full >= 90.0, diagonal=1)
        got = torch.zeros_like(want)
        got[ri, qi] = True
        assert int(sel.sum()) == int(want.sum()) and bool((got == want).all())
        assert float((ani - full[ri, qi]).abs().max()) <= ANI_TOL
        # independent of the kernels: CPU oracle and fp64 GEMM on a 512-row block of the REAL sketches
        rows = slice(1490, 2002)
        for indep, tol in zip(_independent_block(orc, hv, n2, rows), (ANI_TOL, 1e-4)):
            assert float((full[rows] - indep).abs().max()) <= tol
            assert _check_hits_against_block(h, rows, indep, 90.0, True, tol) > 0
        # estimator sanity on real data: member m of cluster 0 vs its root
        est = full[0, 1:100].cpu().numpy()
        assert np.abs(est - 100.0 * (1.0 - 0.001 * np.arange(1, 100))).max() < 1.0
    finally:
        ctx.close()


def test_one_genome_of_more_than_2_to_32_bases():
    """Positions inside one genome are 64-bit (the reference's are usize): a 4.3 Gbp sequence -- a large plant or amphibian
    chromosome set -- is sketched whole, ASCII- and 2-bit-resident, and as three overlapping pieces given as three genomes.
    With k = 31 no sampled k-mer occurs twice in random sequence, so the pieces' hash sets are disjoint: the hash counts and
    the (wrapping int16) hypervectors of the pieces must add up to the whole genome's.  A position truncated to 32 bits
    anywhere on the path would make the whole differ from the pieces, which all stay below 2^32."""
    import hypergen_amd as hg
    dev = torch.device("cuda:0")
    K = 31
    n_bps = (1 << 32) + 3_000_123
    ctx = hg.Context(0)
    ctx.set_stream(torch.cuda.current_stream().cuda_stream)
    try:
        seq = torch.empty(n_bps + 64, dtype=torch.uint8, device=dev)
        lut = torch.tensor(list(b"ACGT"), dtype=torch.uint8, device=dev)
        gen = torch.Generator(device=dev)
        gen.manual_seed(20)
        step = 1 << 28
        for a in range(0, n_bps, step):
            m = min(step, n_bps - a)
            seq[a:a + m] = lut[torch.randint(0, 4, (m,), generator=gen, device=dev, dtype=torch.int32).long()]
        # not-a-base runs: one across the 2^32 boundary, one behind it, lower case in front of it
        seq[(1 << 32) - 5:(1 << 32) + 9] = ord("N")
        seq[(1 << 32) + 1_000_000:(1 << 32) + 1_000_040] = ord("n")
        seq[(1 << 32) - 2_000:(1 << 32) - 1_000] += 32
        torch.cuda.synchronize()
        p = hg.default_params(ksize=K)
        cuts = [0, (1 << 31) + 76, (1 << 32) - 1_000_000, n_bps]  # piece starts are multiples of 4
        offs = np.array([0] + cuts[:-1], np.uint64)
        lens = np.array([n_bps] + [min(n_bps, cuts[i + 1] + K - 1) - cuts[i] for i in range(3)], np.uint64)
        hv = torch.empty((4, D), dtype=torch.int16, device=dev)
        n2 = torch.empty(4, dtype=torch.int32, device=dev)
        nh = torch.empty(4, dtype=torch.int32, device=dev)
        ctx.sketch_batch_dev(seq.data_ptr(), offs, lens, p, hv.data_ptr(), n2.data_ptr(), nh.data_ptr())
        ctx.sync()  # (reads the step's check word, then waits for the stream)
        nhc = nh.cpu().numpy().astype(np.int64)
        assert abs(nhc[0] - n_bps / 1500) < 6 * (n_bps / 1500) ** 0.5  # FracMinHash keeps one k-mer in `scaled`
        assert nhc[0] == nhc[1:].sum(), nhc
        assert torch.equal(hv[1:].int().sum(0).to(torch.int16), hv[0])
        assert torch.equal((hv.int() ** 2).sum(1).int(), n2)
        # the same genome 2-bit resident (hg_pack2 on the device): the same sketch
        blob = torch.empty(hg.lib().hg_pack2_size(n_bps) + 64, dtype=torch.uint8, device=dev)
        ctx.pack2_dev(seq.data_ptr(), n_bps, blob.data_ptr())
        hv_p = torch.empty((1, D), dtype=torch.int16, device=dev)
        n2_p = torch.empty(1, dtype=torch.int32, device=dev)
        nh_p = torch.empty(1, dtype=torch.int32, device=dev)
        ctx.sketch_batch_dev_packed(blob.data_ptr(), np.zeros(1, np.uint64), np.array([n_bps], np.uint64), p, hv_p.data_ptr(),
                                    n2_p.data_ptr(), nh_p.data_ptr())
        ctx.sync()  # (reads the step's check word, then waits for the stream)
        assert int(nh_p[0]) == nhc[0] and int(n2_p[0]) == int(n2[0]) and torch.equal(hv_p[0], hv[0])
    finally:
        ctx.close()


def test_config4_hamming_search_50k_refs_fullsize(orc):
    """BASELINE configs[4] at full size: 50 000 bit-packed D = 16384 references x 1 000 queries.  The hit list is
    compared (a) with the CPU oracle's popcount matrix on a 2 000 x 1 000 slice, (b) for ALL 5*10^7 pairs with an
    independent brute-force evaluation (+-1 bf16 GEMM in torch: hamming = (D - dot) / 2, exact in the f32
    accumulator), and (c) with the same search run as 8 reference shards with global indices (the 8-GPU
    decomposition), which must give the identical hit set."""
    import hypergen_amd as hg
    from hypergen_amd import shard
    dev = torch.device("cuda:0")
    HD, R, Q, MAXD = 16384, 50_000, 1_000, 7_400
    words = HD // 32
    g = torch.Generator(device=dev)
    g.manual_seed(4242)
    refs = torch.randint(-2**31, 2**31 - 1, (R, words), dtype=torch.int32, device=dev, generator=g)
    src = torch.randint(0, R, (Q,), device=dev, generator=g)
    # queries: a source row with a random 0..100 % of its WORDS replaced -> distances from 0 up to the random-pair 8 192
    frac = torch.rand((Q, 1), device=dev, generator=g)
    noise = torch.randint(-2**31, 2**31 - 1, (Q, words), dtype=torch.int32, device=dev, generator=g)
    qry = torch.where(torch.rand((Q, words), device=dev, generator=g) < frac, noise, refs[src])
    ctx = hg.Context(0)
    ctx.set_stream(torch.cuda.current_stream().cuda_stream)
    try:
        cap = 1 << 20
        out = torch.empty(cap * 3, dtype=torch.int32, device=dev)
        n, st = ctx.hamming_search_dev(refs.data_ptr(), R, qry.data_ptr(), Q, HD, MAXD, out.data_ptr(), cap)
        torch.cuda.synchronize()
        assert st == 0 and Q // 2 < n < cap  # ~90 % of the queries keep their source within MAXD
        h = out[: 3 * n].view(n, 3).clone()
        # (b) brute force over all pairs
        def pm1(x):  # bits -> +-1 bf16, LSB first
            sh = torch.arange(32, device=dev, dtype=torch.int32)
            return (((x[:, :, None] >> sh) & 1).to(torch.bfloat16) * 2 - 1).reshape(x.shape[0], -1)
        qf = pm1(qry)
        dist_all = torch.empty((R, Q), dtype=torch.int32, device=dev)
        for r0 in range(0, R, 5000):
            dot = pm1(refs[r0:r0 + 5000]).float() @ qf.float().T  # f32 GEMM: sums of +-1 are exact
            dist_all[r0:r0 + 5000] = ((HD - dot) / 2).round().int()
        want = dist_all <= MAXD
        assert n == int(want.sum())
        got = torch.zeros_like(want)
        got[h[:, 0].long(), h[:, 1].long()] = True
        assert bool((got == want).all())
        assert bool((dist_all[h[:, 0].long(), h[:, 1].long()] == h[:, 2]).all())
        # (a) the CPU oracle on a slice
        sl = orc.hamming_matrix(refs[:2000].cpu().numpy().view(np.uint32), qry.cpu().numpy().view(np.uint32))
        assert np.array_equal(sl, dist_all[:2000].cpu().numpy().astype(np.uint32))
        # (c) 8 reference shards with global indices
        parts = []
        for r in range(8):
            lo, hi = shard.shard_range(R, r, 8)
            m, st = ctx.hamming_search_block_dev(refs[lo:hi].data_ptr(), hi - lo, lo, qry.data_ptr(), Q, 0, HD, MAXD,
                                                 out.data_ptr(), cap)
            assert st == 0
            parts.append(out[: 3 * m].view(m, 3).clone())
        hs = torch.cat(parts)
        key = lambda t: t[:, 0].long() * Q + t[:, 1].long()
        a, b = h[key(h).argsort()], hs[key(hs).argsort()]
        assert a.shape == b.shape and bool((a == b).all())
    finally:
        ctx.close()


def test_device_decode_of_a_10k_sketch_file_image(orc, tmp_path):
    """configs[3]'s input side at full size: 10 000 sketches written as a .sketch file (47 MB), read back as an image,
    uploaded as it is and decoded by hg_hv_unpack_batch_dev -- every row equals the matrix that went in; every 16th record is
    in the reference's non-AVX2 layout (which cannot represent -2^(q-1): those rows are compared with the oracle's decoder)"""
    import bench
    import hypergen_amd as hg
    dev = torch.device("cuda:0")
    n = 10_000
    hv = bench.clustered_hvs(n, 0, dev).cpu().numpy()
    recs = []
    for i in range(n):
        if i % 16 == 7:
            q, pk = hg.hv_pack_naive(hv[i])
        else:
            q, pk = hg.hv_pack(hv[i])
        recs.append(dict(ksize=21, scaled=1500, canonical=True, seed=123, hv_d=4096, hv_quant_bits=q, hv_norm_2=i,
                         file_str="/g/%s%d.fna" % ("y" * (i % 5), i), hv=pk.view(np.int16)))
    path = str(tmp_path / "db.sketch")
    hg.write_sketch_file(path, recs)
    img, meta = hg.read_sketch_file_image(path)
    assert len(meta) == n and img.size > 40_000_000
    lay = [hg.hv_payload_layout(4096, m["hv_quant_bits"], m["payload_bytes"]) for m in meta]
    assert lay.count(hg.PAYLOAD_NAIVE) == n // 16 and min(lay) == 0
    with hg.Context(0) as ctx:
        d_img = torch.from_numpy(img).to(dev)
        out = torch.empty((n, 4096), dtype=torch.int16, device=dev)
        ctx.hv_unpack_batch_dev(d_img.data_ptr(), d_img.numel(), [m["payload_off"] for m in meta], [m["hv_quant_bits"] for m in meta],
                                lay, 4096, out.data_ptr())
        got = out.cpu().numpy()
    bp = np.array(lay) == hg.PAYLOAD_BITPACKER8X
    assert np.array_equal(got[bp], hv[bp])
    for i in np.nonzero(~bp)[0][::25]:
        assert np.array_equal(got[i], orc.unpack_hv_naive(recs[i]["hv"], 4096, recs[i]["hv_quant_bits"])), i
    # (all but the rows holding exactly -2^(q-1) round-trip in the naive layout too)
    assert (got[~bp] == hv[~bp]).mean() > 0.999


def test_dist_of_more_than_2_to_32_pairs():
    """66 000 x 66 000 = 4.36e9 pairs, more than the kernels' 32-bit hit counter could count if every pair were reported:
    hg_dist_dev splits such a call into blocks of reference rows by itself.  The hits must be those of two explicit
    hg_dist_block_dev calls on row blocks that stay below 2^32 pairs each (same global indices, same ANI bits)."""
    import bench
    import hypergen_amd as hg
    dev = torch.device("cuda:0")
    n = 66_000
    assert n * n > 2**32
    hv = bench.clustered_hvs(n, 0, dev)
    n2 = (hv.int() ** 2).sum(1).int()
    cap = 40_000_000
    out = torch.empty(cap * 3, dtype=torch.int32, device=dev)

    def canon(t, k):
        h = t[: 3 * k].view(-1, 3)
        key = (h[:, 0].long() << 32) | (h[:, 1].long() & 0xFFFFFFFF)
        order = torch.argsort(key)
        return key[order].clone(), h[order, 2].clone()
    with hg.Context(0) as c:
        c.set_stream(torch.cuda.current_stream().cuda_stream)
        found, st = c.dist_dev(hv.data_ptr(), n2.data_ptr(), n, hv.data_ptr(), n2.data_ptr(), n, D, 21, False, 85.0, out.data_ptr(), cap)
        torch.cuda.synchronize()
        assert st == 0 and found > n * 100
        k_all, v_all = canon(out, found)
        half = n // 2
        parts, total = [], 0
        for r0, rows in ((0, half), (half, n - half)):
            m, st = c.dist_block_dev(hv[r0:].data_ptr(), n2[r0:].data_ptr(), rows, r0, hv.data_ptr(), n2.data_ptr(), n, 0, D, 21, False, 85.0,
                                     out.data_ptr(), cap)
            torch.cuda.synchronize()
            assert st == 0
            parts.append(canon(out, m))
            total += m
        assert total == found
        k2 = torch.cat([p[0] for p in parts])
        v2 = torch.cat([p[1] for p in parts])
        o = torch.argsort(k2)
        assert torch.equal(k2[o], k_all) and torch.equal(v2[o], v_all)


def test_cli_dist_10k_x_10k_tsv_equals_the_oracles_bytes(tmp_path, orc):
    """BASELINE configs[3] through the drop-in surface: `hyper-gen dist -r A.sketch -q B.sketch` on 10 000 x 10 000 sketches
    must write, byte for byte, the TSV the reference writes -- ANI values from the ORACLE's formula (orc_ani_from_dot: the
    host's logf, float32 operations in src/dist.rs:153-160's order) on exact dot products, pairs in dist.rs:243-265's
    enumeration order, dump_ani_file's stable-sort-then-reverse order (src/utils.rs:260-308), "{:.3}" text.
    The 10^8 dot products come from an fp64 torch GEMM (exact: |dot| << 2^53; 4 * 10^11 scalar MACs are out of the oracle's
    reach in a test); a 256-row band of them is checked against the oracle's own scalar dot (orc_ani_matrix)."""
    import subprocess
    import sys
    sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
    import bench
    import hypergen_amd as hg
    dev = torch.device("cuda:0")
    n = 10000
    a = bench.clustered_hvs(n, 0, dev)
    b = bench.clustered_hvs(n, 0, dev, salt=1)
    n2a, n2b = (a.int() ** 2).sum(1).int(), (b.int() ** 2).sum(1).int()
    # expected ANI of every pair: exact dots (fp64 GEMM in row blocks) -> the oracle's formula on the host
    ani = np.empty((n, n), np.float32)
    bd, nb_ = b.double(), n2b.cpu().numpy()
    for r0 in range(0, n, 2000):
        dots = (a[r0:r0 + 2000].double() @ bd.T).round().to(torch.int32).cpu().numpy()
        nr = np.repeat(n2a[r0:r0 + 2000].cpu().numpy(), n)
        ani[r0:r0 + 2000] = orc.ani_from_dots(dots.ravel(), nr, np.tile(nb_, dots.shape[0]), 21).reshape(dots.shape)
    del bd
    band = slice(4900, 5156)
    ah, bh = a.cpu().numpy(), b.cpu().numpy()
    assert np.array_equal(ani[band], orc.ani_matrix(ah[band], n2a.cpu().numpy()[band], bh, nb_, 21))
    paths, names = [], []
    for tag, hv, n2 in (("a", ah, n2a.cpu().numpy()), ("b", bh, nb_)):
        recs = []
        for i in range(n):
            q, pk = hg.hv_pack(hv[i])
            recs.append(dict(ksize=21, scaled=1500, canonical=True, seed=123, hv_d=4096, hv_quant_bits=q, hv_norm_2=int(n2[i]),
                             file_str="/data/%s/genome_%05d.fna" % (tag, i), hv=pk.view(np.int16)))
        p = str(tmp_path / (tag + ".sketch"))
        hg.write_sketch_file(p, recs)
        paths.append(p)
        names.append([r["file_str"] for r in recs])
    del a, b
    torch.cuda.empty_cache()
    # the expected text (the pairs >= 85: ~1.3 M lines)
    ii, jj = np.nonzero(ani >= np.float32(85.0))  # row-major = enumeration order
    v = ani[ii, jj]
    order = np.argsort(v, kind="stable")[::-1]
    want = "".join("%s\t%s\t%.3f\n" % (names[0][ii[t]], names[1][jj[t]], float(v[t])) for t in order)
    tsv = str(tmp_path / "ani.tsv")
    r = subprocess.run([hg.CLI_PATH, "dist", "-r", paths[0], "-q", paths[1], "-o", tsv, "-a", "85"], capture_output=True, text=True)
    assert r.returncode == 0, r.stderr
    got = open(tsv).read()
    assert len(order) > 1_000_000
    if got != want:  # say where, not just that
        gl, wl = got.splitlines(), want.splitlines()
        first = next((t for t in range(min(len(gl), len(wl))) if gl[t] != wl[t]), min(len(gl), len(wl)))
        raise AssertionError("TSV differs: %d vs %d lines, first difference at line %d: %r vs %r" % (
            len(gl), len(wl), first, gl[first] if first < len(gl) else None, wl[first] if first < len(wl) else None))
