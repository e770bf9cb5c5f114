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
a = ap.parse_args()
dev = torch.device("cuda:0")
ctx = hg.Context(0)
ctx.set_stream(torch.cuda.current_stream().cuda_stream)
hv = bench.clustered_hvs(a.n, 0, dev)
n2 = (hv.int() ** 2).sum(1).int()
cap = max(1 << 20, a.n * a.n // 20)
hits = torch.empty(cap * 3, dtype=torch.int32, device=dev)
ctx.enable_timing(True)
for r in range(a.reps + 1):
    ctx.timings()
    found, _ = ctx.dist_dev(hv.data_ptr(), n2.data_ptr(), a.n, hv.data_ptr(), n2.data_ptr(), a.n, 4096, 21, False,
                            a.th, hits.data_ptr(), cap)
    torch.cuda.synchronize()
    t = ctx.timings()
    if r:
        ms = t["dist"][0]
        print("dist %dx%d: gemm %.3f ms = %.1f TFLOP/s, prep %.3f ms, hits %d" % (
            a.n, a.n, ms, a.n * a.n * 8192 / ms / 1e9, t["dist_prep"][0], found))
