#!/usr/bin/env python3
"""dist-only timing / profiling target: R x Q clustered synthetic HVs, thresholded output."""
import argparse
import os
import sys

import torch

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import hypergen_amd as hg  # noqa: E402
import bench  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--n", type=int, default=10000)
ap.add_argument("--reps", type=int, default=3)
ap.add_argument("--th", type=float, default=85.0)
ap.add_argument("--nhash", type=int, default=3333, help="hashes per sketch (3333 = 5 Mbp at scaled 1500); > 4096 needs two exact f32 windows at D = 4096")
ap.add_argument("--sets", default="same", choices=("same", "two", "sym"),
                help="same: one buffer on both sides (BASELINE.md's 10 000 HVs against themselves); two: R and Q are different "
                     "members of the same clusters in different buffers (both prepasses, two operand matrices); sym: one set, "
                     "symmetric = 1 (the reference's path_r == path_q case, src/dist.rs:13,243-265)")
a = ap.parse_args()
dev = torch.device("cuda:0")
ctx = hg.Context(0)
ctx.set_stream(torch.cuda.current_stream().cuda_stream)
hv = bench.clustered_hvs(a.n, 0, dev, n=a.nhash)
n2 = (hv.int() ** 2).sum(1).int()
qv, qn2 = hv, n2
if a.sets == "two":
    qv = bench.clustered_hvs(a.n, 0, dev, n=a.nhash, salt=1)
    qn2 = (qv.int() ** 2).sum(1).int()
SYM = a.sets == "sym"
PAIRS = a.n * (a.n - 1) // 2 if SYM else a.n * a.n
cap = max(1 << 20, a.n * a.n // 20)
hits = torch.empty(cap * 3, dtype=torch.int32, device=dev)
ctx.enable_timing(True)
# HG_DIST_VARIANTS = comma list of "<tile>[:<path>[:<order>]]" (hg_ctx_set_debug keys dist_tile / dist_path / dist_order), e.g. ",:f16,wide:i8,::legacy"
variants = os.environ.get("HG_DIST_VARIANTS", "").split(",")
res = {v: [] for v in variants}
for r in range(a.reps + 1):
    for v in variants:  # interleaved so that clock / thermal drift hits every variant alike
        ctx.set_debug("dist_tile", v.split(":")[0])
        ctx.set_debug("dist_path", v.split(":")[1] if ":" in v else "")
        ctx.set_debug("dist_order", v.split(":")[2] if v.count(":") > 1 else "")  # "" | "plain" | "legacy"
        ctx.timings()
        found, _ = ctx.dist_dev(hv.data_ptr(), n2.data_ptr(), a.n, qv.data_ptr(), qn2.data_ptr(), a.n, 4096, 21, SYM,
                                a.th, hits.data_ptr(), cap)
        torch.cuda.synchronize()
        t = ctx.timings()
        if r:
            res[v].append((t["dist"][0], t["dist_prep"][0], found))
for v in variants:
    ms = sorted(x[0] for x in res[v])
    med, best = ms[len(ms) // 2], ms[0]
    print("dist %dx%d %s [%s]: gemm median %.3f ms = %.1f TFLOP/s (best %.3f ms = %.1f), prep %.3f ms, hits %d" % (
        a.n, a.n, a.sets, v or "default", med, PAIRS * 8192 / med / 1e9, best, PAIRS * 8192 / best / 1e9,
        sorted(x[1] for x in res[v])[len(ms) // 2], res[v][0][2]))
if os.environ.get("HG_DIST_WALL"):
    import time
    for _ in range(3):
        ctx.dist_dev(hv.data_ptr(), n2.data_ptr(), a.n, qv.data_ptr(), qn2.data_ptr(), a.n, 4096, 21, SYM, a.th,
                     hits.data_ptr(), cap)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(20):
        ctx.dist_dev(hv.data_ptr(), n2.data_ptr(), a.n, qv.data_ptr(), qn2.data_ptr(), a.n, 4096, 21, SYM, a.th,
                     hits.data_ptr(), cap)
    torch.cuda.synchronize()
    print("wall per call: %.3f ms" % ((time.perf_counter() - t0) / 20 * 1e3))
