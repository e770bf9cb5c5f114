#!/usr/bin/env python3
"""Bit-packed Hamming search over a ladder of shapes, dimensions and thresholds: wall time per call and pairs/s -- looking for
parameters that fall off the 50 000 x 10 000 x 16 384 rate.  Needs a GPU."""
import os
import sys
import time

import torch

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import hypergen_amd as hg  # noqa: E402

dev = torch.device("cuda:0")
ctx = hg.Context(0)
ctx.set_stream(torch.cuda.current_stream().cuda_stream)
g = torch.Generator(device=dev)
g.manual_seed(5)
cap = 40_000_000
hits = torch.empty(cap * 3, dtype=torch.int32, device=dev)
base = None
for D, R, Q, frac in ((16384, 50_000, 10_000, 0.40), (16384, 50_000, 1_000, 0.40), (16384, 50_000, 10, 0.40), (16384, 1_000_000, 10, 0.40),
                      (16384, 1_000, 1_000, 0.40), (16384, 50_000, 10_000, 0.49), (16384, 20_000, 2_000, 0.51),
                      (4096, 100_000, 10_000, 0.40), (8192, 50_000, 10_000, 0.40), (16320, 50_000, 10_000, 0.40), (32768, 50_000, 5_000, 0.40)):
    words = (D + 63) // 64
    rb = torch.randint(-2**62, 2**62, (R, words), generator=g, device=dev, dtype=torch.int64)
    qb = rb[:Q].clone()
    qb[:, 0] ^= 0x5555  # every query is a near copy of a reference: Q planted hits
    max_dist = int(D * frac)
    for _ in range(2):
        n, st = ctx.hamming_search_dev(rb.data_ptr(), R, qb.data_ptr(), Q, D, max_dist, hits.data_ptr(), cap)
    torch.cuda.synchronize()
    reps = 5
    t0 = time.perf_counter()
    for _ in range(reps):
        n, st = ctx.hamming_search_dev(rb.data_ptr(), R, qb.data_ptr(), Q, D, max_dist, hits.data_ptr(), cap)
    torch.cuda.synchronize()
    ms = (time.perf_counter() - t0) / reps * 1e3
    bitops = R * Q * D / ms / 1e9
    if base is None:
        base = bitops
    print("hamming D=%5d %8d x %6d, max_dist %.2f D: %9.3f ms, %7.0f M pairs/s, %6.1f T bit-pairs/s (%.2f of the first), %d hits, status %d" % (
        D, R, Q, frac, ms, R * Q / ms / 1e3, bitops, bitops / base, n, st), flush=True)
    del rb, qb
