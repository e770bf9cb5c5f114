#!/usr/bin/env python3
"""Where the dist kernel's time goes beyond its main loop: every tile of the 10 000 x 10 000 clustered comparison split into
main loop / phase 0 (masks) / append / phase 2 / reservation / hit write / closing barrier, from in-kernel s_memtime stamps
of ALL 1 280 workgroups (development build: tools/build_variant.sh stamps hg_dist_kernels.hip -DHG_DIST_STAMPS; run with
HYPERGEN_LIB=tools/_exp_lib_stamps.so).  Once without candidates (ani_th 101) and once with the bench's 1.29 M hits (85):
the difference of the two, phase by phase and summed along the tiles each CU runs, is the kernel-time gap between them.
Also: how many tiles are dense, which tiles end last, how long the last CU of each XCD runs alone."""
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
PH = ["main loop", "norms staged + masks (phase 0)", "append", "phase 2 (candidates evaluated)", "reservation", "hits written",
      "closing barrier", "rest of the epilogue"]


def run(th, reps=300):
    for _ in range(reps):
        ctx.dist_dev(hv.data_ptr(), n2.data_ptr(), n, hv.data_ptr(), n2.data_ptr(), n, 4096, 21, False, th, hits.data_ptr(), cap)
    torch.cuda.synchronize()
    best = 1e9
    for _ in range(9):
        ctx.enable_timing(True)
        ctx.timings()
        for _ in range(20):
            found, _ = ctx.dist_dev(hv.data_ptr(), n2.data_ptr(), n, hv.data_ptr(), n2.data_ptr(), n, 4096, 21, False, th, hits.data_ptr(), cap)
        torch.cuda.synchronize()
        tm = ctx.timings()
        ctx.enable_timing(False)
        best = min(best, tm["dist"][0] / tm["dist"][1])
    al = np.zeros((2048, 16), np.uint64)
    assert hg.lib().hg_debug_dist_tile_all(C.c_void_p(al.ctypes.data)) == 0
    al = al.astype(np.int64)
    ok = (al[:, 2] > al[:, 0]) & (al[:, 2] - al[:, 0] < 4000000) & (al[:, 1] > al[:, 0])
    idx = np.nonzero(ok)[0]
    a = al[idx]
    tot = a[:, 2] - a[:, 0]
    main = a[:, 1] - a[:, 0]
    hot = a[:, 7] > 0  # reached the append / a flush
    p0 = np.where(a[:, 6] > 0, a[:, 6] - a[:, 1], a[:, 2] - a[:, 1])
    app = np.where(hot, a[:, 7] - a[:, 6], 0)
    p2, rsv, wr, cb = a[:, 8], a[:, 9], a[:, 10], a[:, 11]
    rest = tot - main - p0 - app - p2 - rsv - wr - cb
    return dict(ms=best, found=found, idx=idx, a=a, tot=tot, parts=np.stack([main, p0, app, p2, rsv, wr, cb, rest], 1), hot=hot,
                cand=a[:, 3], flushes=a[:, 12], nh=a[:, 13], kernel=ctx.last_kernel("dist"))


cold, warm = run(101.0), run(85.0)
print("kernel: %s" % warm["kernel"])
print("kernel time, best of 9 x 20 launches (stamped build): no candidates %.4f ms, %d hits %.4f ms -> gap %.1f us" % (
    cold["ms"], warm["found"], warm["ms"], (warm["ms"] - cold["ms"]) * 1e3))
# shader clock from the span of the stamped tiles against the kernel time is not available per XCD; use ticks -> us through
# the kernel itself: 5 tiles per CU, kernel time = the slowest CU's tiles
for name, r in (("no candidates", cold), ("1.29 M hits", warm)):
    print("\n%s: %d tiles stamped, %d reached the lists, candidates %d, hits %d" % (name, r["idx"].size, int(r["hot"].sum()), int(r["cand"].sum()), int(r["nh"].sum())))
    print("   %-34s %10s %10s %10s" % ("ticks per tile", "mean", "median", "max"))
    for k, nm in enumerate(PH):
        v = r["parts"][:, k]
        print("   %-34s %10.0f %10.0f %10.0f" % (nm, v.mean(), np.median(v), v.max()))
    print("   %-34s %10.0f %10.0f %10.0f" % ("tile total", r["tot"].mean(), np.median(r["tot"]), r["tot"].max()))
d = warm["parts"].mean(0) - cold["parts"].mean(0)
tile_ticks = cold["tot"].mean()
tiles_per_cu = warm["idx"].size / 256.0
ticks_per_us = tile_ticks * tiles_per_cu / (cold["ms"] * 1e3)  # the cold kernel is its CUs' tiles back to back (entry/exit aside)
print("\nper-tile difference (1.29 M hits - no candidates), mean ticks, and what %.2f tiles per CU make of it at %.0f ticks/us:" % (tiles_per_cu, ticks_per_us))
for k, nm in enumerate(PH):
    print("   %-34s %+9.0f ticks  = %+6.1f us per CU" % (nm, d[k], d[k] * tiles_per_cu / ticks_per_us))
print("   %-34s %+9.0f ticks  = %+6.1f us per CU   (measured gap %.1f us)" % ("sum", d.sum(), d.sum() * tiles_per_cu / ticks_per_us, (warm["ms"] - cold["ms"]) * 1e3))
# dense tiles and the tail
r = warm
dense = r["flushes"] > 1
print("\ntiles by candidates: none %d, 1-2 000 %d, 2 000-10 000 %d, > 10 000 %d; tiles with more than one flush (dense): %d (%s flushes)" % (
    int((r["cand"] == 0).sum()), int(((r["cand"] > 0) & (r["cand"] <= 2000)).sum()), int(((r["cand"] > 2000) & (r["cand"] <= 10000)).sum()),
    int((r["cand"] > 10000).sum()), int(dense.sum()), sorted(set(r["flushes"][dense].tolist()))))
if dense.any():
    print("   dense tiles: total %.0f ticks mean (%.2f x an ordinary hot tile's %.0f), slots %s" % (
        r["tot"][dense].mean(), r["tot"][dense].mean() / r["tot"][r["hot"] & ~dense].mean(), r["tot"][r["hot"] & ~dense].mean(),
        r["idx"][dense][:12].tolist()))
# per CU: XCC id (bits 0-3 of column 4), HW_ID >> 8: CU id bits 8-11, SH 12, SE 13-15 (gfx9 HW_ID layout)
hw = r["a"][:, 4] >> 8
cu_key = (r["a"][:, 4] & 15) * 4096 + ((hw >> 8) & 15) + 16 * ((hw >> 12) & 1) + 32 * ((hw >> 13) & 7)
for name, rr in (("no candidates", cold), ("1.29 M hits", warm)):
    hw_ = rr["a"][:, 4] >> 8
    key = (rr["a"][:, 4] & 15) * 4096 + ((hw_ >> 8) & 15) + 16 * ((hw_ >> 12) & 1) + 32 * ((hw_ >> 13) & 7)
    cus = np.unique(key)
    busy = np.array([rr["tot"][key == c].sum() for c in cus])
    ntile = np.array([(key == c).sum() for c in cus])
    span = np.array([rr["a"][key == c, 2].max() - rr["a"][key == c, 0].min() for c in cus])
    print("%s: %d CUs seen, tiles per CU %d..%d, busy ticks per CU mean %.0f max %.0f (max / mean %.3f), span mean %.0f max %.0f" % (
        name, cus.size, ntile.min(), ntile.max(), busy.mean(), busy.max(), busy.max() / busy.mean(), span.mean(), span.max()))
    # tail per XCD: the last tile end against the mean of the CUs' last ends
    for x in range(8):
        m = (rr["a"][:, 4] & 15) == x
        if not m.any():
            continue
        ends = np.array([rr["a"][(key == c) & m, 2].max() for c in np.unique(key[m])])
        t0 = rr["a"][m, 0].min()
        print("   XCD %d: %3d tiles on %2d CUs, last CU ends at %7d ticks, mean CU end %7d, the last CU runs alone for %5d ticks (%.1f us)" % (
            x, int(m.sum()), ends.size, ends.max() - t0, ends.mean() - t0, ends.max() - np.sort(ends)[-2] if ends.size > 1 else 0,
            (ends.max() - np.sort(ends)[-2]) / ticks_per_us if ends.size > 1 else 0))
# which tiles end last in the hot run
order = np.argsort(r["a"][:, 2] - r["a"][:, 0].min())[-8:]
print("last finishers of the hot run (slot, xcd, candidates, flushes, tile ticks):", [(int(r["idx"][i]), int(r["a"][i, 4] & 15), int(r["cand"][i]), int(r["flushes"][i]), int(r["tot"][i])) for i in order])
