#!/usr/bin/env python3
"""dist at other hypervector dimensions than 4096: thresholded hits against the full-matrix mode + timing."""
import sys, os, time
sys.path.insert(0, ".")
import torch, numpy as np
import hypergen_amd as hg, bench
dev = torch.device("cuda:0")
ctx = hg.Context(0)
ctx.set_stream(torch.cuda.current_stream().cuda_stream)
n = 6000
for D in (256, 1024, 2048, 8192, 16384, 4160):
    bench.HV_D = D
    hv = bench.clustered_hvs(n, 0, dev)
    n2 = (hv.int() ** 2).sum(1).int()
    full = torch.empty((n, n), dtype=torch.float32, device=dev)
    ctx.dist_full_dev(hv.data_ptr(), n2.data_ptr(), n, hv.data_ptr(), n2.data_ptr(), n, D, 21, full.data_ptr())
    torch.cuda.synchronize()
    cap = 6_000_000
    hits = torch.empty(cap * 3, dtype=torch.int32, device=dev)
    th = 90.0
    for rep in range(3):
        torch.cuda.synchronize(); t = time.time()
        found, st = ctx.dist_dev(hv.data_ptr(), n2.data_ptr(), n, hv.data_ptr(), n2.data_ptr(), n, D, 21, False, th, hits.data_ptr(), cap)
        torch.cuda.synchronize(); dt = time.time() - t
    want = int((full >= th).sum())
    h = hits[: found * 3].view(found, 3)
    err = float((h[:, 2].view(torch.float32) - full[h[:, 0].long(), h[:, 1].long()]).abs().max()) if found else 0.0
    print("D=%5d: %.3f ms (%.0f TFLOP/s), hits %d, full>=th %d, max |dANI| %.2g %s" % (
        D, dt * 1e3, n * n * 2.0 * D / dt / 1e12, found, want, err, "OK" if found == want and err <= 1e-4 else "MISMATCH"))
