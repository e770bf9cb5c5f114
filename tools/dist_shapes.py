#!/usr/bin/env python3
"""dist over a ladder of R x Q shapes (square, skinny, wide, large): wall time per call, pairs/s and what the same pair count
would take at the 10 000 x 10 000 rate -- looking for shapes that fall off it.  Needs a GPU."""
import os
import sys
import time

import torch

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import hypergen_amd as hg  # noqa: E402
import bench  # noqa: E402

dev = torch.device("cuda:0")
ctx = hg.Context(0)
ctx.set_stream(torch.cuda.current_stream().cuda_stream)
big = bench.clustered_hvs(100_000, 0, dev)
big_n2 = (big.int() ** 2).sum(1).int()
cap = 60_000_000
hits = torch.empty(cap * 3, dtype=torch.int32, device=dev)
base = None
for R, Q in ((10_000, 10_000), (1_000, 1_000), (100, 100), (100_000, 100), (100, 100_000), (100_000, 1), (1, 100_000), (100_000, 10), (100_000, 16),
             (100_000, 1_000), (1_000, 100_000), (30_000, 30_000), (100_000, 100_000)):
    r, rn = big[:R], big_n2[:R]
    q, qn = big[100_000 - Q:], big_n2[100_000 - Q:]
    reps = 3 if R * Q > 2e9 else 10
    for _ in range(2):
        found, st = ctx.dist_dev(r.data_ptr(), rn.data_ptr(), R, q.data_ptr(), qn.data_ptr(), Q, 4096, 21, False, 85.0, hits.data_ptr(), cap)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(reps):
        found, st = ctx.dist_dev(r.data_ptr(), rn.data_ptr(), R, q.data_ptr(), qn.data_ptr(), Q, 4096, 21, False, 85.0, hits.data_ptr(), cap)
    torch.cuda.synchronize()
    ms = (time.perf_counter() - t0) / reps * 1e3
    if base is None:
        base = R * Q / ms
    print("dist %7d x %7d: %9.3f ms per call, %8.0f M pairs/s (%.2f of the 10 000^2 rate), %d hits, status %d" % (
        R, Q, ms, R * Q / ms / 1e3, R * Q / ms / base, found, st), flush=True)
