#!/usr/bin/env python3
"""Per-CU timelines of one launch of the dist GEMM from the whole-tile stamps (development build: tools/build_variant.sh
stamps hg_dist_kernels.hip -DHG_DIST_STAMPS; HYPERGEN_LIB=tools/_exp_lib_stamps.so): which workgroups ran on which CU
(HW_ID), when each started and ended, the gaps between consecutive tiles of a CU, rounds per CU and XCD."""
import ctypes as C, os, sys
import numpy as np, torch
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import hypergen_amd as hg, bench
n = int(sys.argv[1]) if len(sys.argv) > 1 else 10000
dev = torch.device("cuda:0")
ctx = hg.Context(0)
ctx.set_stream(torch.cuda.current_stream().cuda_stream)
for kv in os.environ.get("HG_DEBUG", "").split(","):  # e.g. HG_DEBUG=dist_defer=off,dist_order=legacy
    if "=" in kv:
        ctx.set_debug(*kv.split("=", 1))
hv = bench.clustered_hvs(n, 0, dev)
n2 = (hv.int() ** 2).sum(1).int()
cap = 1 << 23
hits = torch.empty(cap * 3, dtype=torch.int32, device=dev)
for th in (101.0, 85.0):
    for _ in range(100):
        ctx.dist_dev(hv.data_ptr(), n2.data_ptr(), n, hv.data_ptr(), n2.data_ptr(), n, 4096, 21, False, th, hits.data_ptr(), cap)
    torch.cuda.synchronize()
    al = np.zeros((2048, 5), np.uint64)
    assert hg.lib().hg_debug_dist_tile_all(C.c_void_p(al.ctypes.data)) == 0
    al = al.astype(np.int64)
    ok = (al[:, 2] > al[:, 0]) & (al[:, 2] - al[:, 0] < 4000000) & (al[:, 1] > al[:, 0])
    idx = np.nonzero(ok)[0]
    xcc = al[:, 4] & 0xFF
    hw = al[:, 4] >> 8
    cu = (hw >> 8) & 0xF
    sh = (hw >> 12) & 1
    se = (hw >> 13) & 7
    print("th %.0f kernel %s: %d tiles stamped, tile mean %.0f ticks" % (th, ctx.last_kernel("dist"), idx.size, (al[idx, 2] - al[idx, 0]).mean()))
    dur = al[idx, 2] - al[idx, 0]
    print("  tile duration: min %d  median %d  p90 %d  p99 %d  max %d; slots 0..39 (first diagonal tiles) median %d max %d; 40..79 median %d" % (
        dur.min(), np.median(dur), np.percentile(dur, 90), np.percentile(dur, 99), dur.max(),
        np.median(al[:40, 2] - al[:40, 0]), (al[:40, 2] - al[:40, 0]).max(), np.median(al[40:80, 2] - al[40:80, 0])))
    spans = []
    for x in range(8):
        m = idx[xcc[idx] == x]
        key = se[m] * 100 + sh[m] * 16 + cu[m]
        for k in np.unique(key):
            w = m[key == k]
            spans.append(al[w, 2].max() - al[w, 0].min())  # (one CU, one counter)
    spans = np.array(spans)
    print("  per CU, first tile's entry -> last tile's end: min %d  median %d  p90 %d  max %d" % (spans.min(), np.median(spans), np.percentile(spans, 90), spans.max()))
    for x in range(8):
        m = idx[xcc[idx] == x]
        t0 = al[m, 0].min()
        key = se[m] * 100 + sh[m] * 16 + cu[m]
        cus = np.unique(key)
        rounds, gaps, firsts, lasts = [], [], [], []
        for k in cus:
            w = m[key == k]
            w = w[np.argsort(al[w, 0])]
            rounds.append(w.size)
            firsts.append(al[w[0], 0] - t0)
            lasts.append(al[w[-1], 2] - t0)
            gaps += list(al[w[1:], 0] - al[w[:-1], 2])
        gaps = np.array(gaps)
        print("  XCD %d: %3d tiles on %2d CUs; tiles per CU min %d max %d (histogram %s); first start spread %6d; span %7d; CU end min %7d max %7d; "
              "gap between tiles of a CU mean %5.0f median %5.0f max %6d" % (
                  x, m.size, cus.size, min(rounds), max(rounds), np.bincount(rounds).tolist(), max(firsts), al[m, 2].max() - t0, min(lasts), max(lasts),
                  gaps.mean() if gaps.size else 0, np.median(gaps) if gaps.size else 0, gaps.max() if gaps.size else 0))
        if x == 0:
            for k in cus[:6]:
                w = m[key == k]
                w = w[np.argsort(al[w, 0])]
                print("     CU %3d: " % k + "  ".join("[wg %4d %7d..%7d]" % (i, al[i, 0] - t0, al[i, 2] - t0) for i in w))
