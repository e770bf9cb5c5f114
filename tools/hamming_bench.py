#!/usr/bin/env python3
"""BASELINE configs[4] on one GPU: 50 000 ref x 1 000 query bit-packed D=16384 HVs, Hamming search."""
import os
import sys

import torch

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import hypergen_amd as hg  # noqa: E402

R, Q, D = 50000, 1000, 16384
dev = torch.device("cuda:0")
ctx = hg.Context(0)
ctx.set_stream(torch.cuda.current_stream().cuda_stream)
g = torch.Generator(device=dev)
g.manual_seed(5)
ref = torch.randint(-2**31, 2**31 - 1, (R, D // 32), dtype=torch.int32, device=dev, generator=g)
qry = ref[:Q].clone()
flip = torch.randint(0, 32, (Q, D // 32), device=dev, generator=g)
qry ^= (1 << flip.clamp(max=30)).int()  # one flipped bit per word: distance 512 to its own ref
cap = 1 << 22
hits = torch.empty(cap * 3, dtype=torch.int32, device=dev)
ctx.enable_timing(True)
for rep in range(4):
    ctx.timings()
    n, st = ctx.hamming_search_dev(ref.data_ptr(), R, qry.data_ptr(), Q, D, 2000, hits.data_ptr(), cap)
    torch.cuda.synchronize()
    ms = ctx.timings()["dist"][0]
    if rep:
        print("hamming %d x %d, D=%d: %.3f ms -> %.1f M pairs/s, %.2f T(xor+popc word-ops)/s, hits %d" % (
            R, Q, D, ms, R * Q / ms / 1e3, R * Q * (D / 32) / ms / 1e9, n))
