#!/usr/bin/env python3
"""Where a tile of the dist GEMM spends its time, from in-kernel s_memtime stamps (development build:
tools/build_variant.sh stamps hg_dist_kernels.hip -DHG_DIST_STAMPS; run with HYPERGEN_LIB=tools/_exp_lib_stamps.so).
10 000 x 10 000 clustered HVs, once without candidates (ani_th 101) and once with (85): the parts of a tile (workgroups
512..527, all waves), the inside of the last flush, the main loop's shader clock (s_memtime against the 100 MHz
s_memrealtime), and over all 1 280 tiles the epilogue's ticks as a linear function of the tile's candidates."""
import ctypes as C, os, sys
import numpy as np, torch
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import hypergen_amd as hg, bench
n = 10000
dev = torch.device("cuda:0")
ctx = hg.Context(0)
ctx.set_stream(torch.cuda.current_stream().cuda_stream)
hv = bench.clustered_hvs(n, 0, dev)
n2 = (hv.int() ** 2).sum(1).int()
cap = 1 << 23
hits = torch.empty(cap * 3, dtype=torch.int32, device=dev)
names = ["entry -> stage 0 landed", "main loop", "norms staged", "accumulator sweep", "final flush"]
for th in (101.0, 85.0):
    for _ in range(200):
        ctx.dist_dev(hv.data_ptr(), n2.data_ptr(), n, hv.data_ptr(), n2.data_ptr(), n, 4096, 21, False, th, hits.data_ptr(), cap)
    torch.cuda.synchronize()
    st = np.zeros((16, 8, 10), np.uint64)
    assert hg.lib().hg_debug_dist_tile_stamps(C.c_void_p(st.ctypes.data)) == 0
    st = st.astype(np.int64)
    d = np.diff(st[:, :, :6], axis=2)
    fl = st[:, :, 6:9] - st[:, :, 4:5]
    print("   inside the last flush (from its start): candidates evaluated %.0f, range reserved %.0f, hits written %.0f" % tuple(fl.mean(axis=(0, 1))))
    print("th %.0f kernel %s: tile total %.0f ticks" % (th, ctx.last_kernel("dist"), (st[:, :, 5] - st[:, :, 0]).mean()))
    for i, nm in enumerate(names):
        print("   %-26s mean %8.0f  min %8.0f  max %8.0f" % (nm, d[:, :, i].mean(), d[:, :, i].min(), d[:, :, i].max()))
    print("   per-wg flush:", (st[:, 0, 5] - st[:, 0, 4]).tolist())
    rl = np.zeros((2048, 2), np.uint64)
    assert hg.lib().hg_debug_dist_tile_real(C.c_void_p(rl.ctypes.data)) == 0
    rl = rl.astype(np.int64)[512:528]
    ml = st[:, 0, 2] - st[:, 0, 1]
    print("   main loop: %.0f shader ticks in %.0f ticks of the 100 MHz clock = %.0f MHz" % (ml.mean(), (rl[:, 1] - rl[:, 0]).mean(), ml.mean() / (rl[:, 1] - rl[:, 0]).mean() * 100))
    al = np.zeros((2048, 16), np.uint64)
    assert hg.lib().hg_debug_dist_tile_all(C.c_void_p(al.ctypes.data)) == 0
    al = al.astype(np.int64)[:1440]
    ok = (al[:, 2] > al[:, 0]) & (al[:, 2] - al[:, 0] < 2000000) & (al[:, 1] > al[:, 0])  # (slots that returned at once keep stale stamps)
    t0 = al[ok, 0].min()
    tot = (al[ok, 2] - al[ok, 0]); epi = al[ok, 2] - al[ok, 1]; cand = al[ok, 3]
    print("   %d tiles stamped; kernel span %.0f ticks; tile total mean %.0f; epilogue mean %.0f (cold tiles %.0f)" % (
        ok.sum(), al[ok, 2].max() - t0, tot.mean(), epi.mean(), epi[cand == 0].mean() if (cand == 0).any() else -1))
    hot = cand > 0
    if hot.any():
        A = np.stack([cand[hot], np.ones(hot.sum())], 1).astype(np.float64)
        k, b = np.linalg.lstsq(A, epi[hot].astype(np.float64), rcond=None)[0]
        print("   hot tiles %d, candidates %d (max %d per tile); epilogue ticks ~ %.2f * candidates + %.0f; max epilogue %.0f" % (
            hot.sum(), cand.sum(), cand.max(), k, b, epi.max()))
        # when do tiles end, relative to the kernel's end: the last 10 finishing tiles
        order = np.argsort(al[ok, 2])[-10:]
        idx = np.nonzero(ok)[0][order]
        print("   last finishers (wg, xcd, candidates, end before kernel end):", [(int(i), int(i % 8), int(al[i, 3]), int(al[ok, 2].max() - al[i, 2])) for i in idx])
        xc = [(int(x), int(cand[(np.nonzero(ok)[0] % 8) == x].sum()), int((al[ok, 2][(np.nonzero(ok)[0] % 8) == x]).max() - t0)) for x in range(8)]
        print("   per XCD (xcd, candidates, last end):", xc)
    # per XCD (s_memtime is not synchronised between XCDs): span of its tiles against the sum of their durations / 32 CUs
    idx = np.nonzero(ok)[0]
    print("   workgroups whose XCC id is not blockIdx %% 8: %d of %d" % ((al[idx, 4] != idx % 8).sum(), idx.size))
    for x in range(8):
        m = al[idx, 4] == x
        a = al[idx[m]]
        span = a[:, 2].max() - a[:, 0].min()
        work = (a[:, 2] - a[:, 0]).sum() / 32.0
        print("   XCD %d: %3d tiles, span %7d ticks, work / 32 CUs %7d (%.0f %%), candidates %d" % (x, m.sum(), span, work, 100.0 * work / span, a[:, 3].sum()))
