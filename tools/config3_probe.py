#!/usr/bin/env python3
"""BASELINE configs[2]/[3] on ONE GPU: 10 000 synthetic 5 Mbp genomes resident in HBM (50 GB) sketched in one
call, then the 10 000 x 10 000 ANI matrix of the sketches, thresholded.  Development / validation aid."""
import sys, time
sys.path.insert(0, ".")
import numpy as np
import torch
import hypergen_amd as hg
from oracle import oracle as orc

dev = torch.device("cuda:0")
ctx = hg.Context(0)
ctx.set_stream(torch.cuda.current_stream().cuda_stream)
N, L, D = 10000, 5_000_000, 4096
stride = (L + 1 + 15) // 16 * 16
seq = torch.empty(N * stride + 64, dtype=torch.uint8, device=dev)
t = time.time(); ctx.synth_genomes_dev(0, N, L, stride, seq.data_ptr()); torch.cuda.synchronize()
print("synth %d genomes (%.1f GB) in %.2f s" % (N, N * stride / 1e9, time.time() - t))
offs = np.arange(N, dtype=np.uint64) * stride
lens = np.full(N, L + 1, np.uint64)
p = hg.default_params()
hv = torch.empty((N, D), dtype=torch.int16, device=dev)
n2 = torch.empty(N, dtype=torch.int32, device=dev)
nh = torch.empty(N, dtype=torch.int32, device=dev)
for rep in range(2):
    torch.cuda.synchronize(); t = time.time()
    ctx.sketch_batch_dev(seq.data_ptr(), offs, lens, p, hv.data_ptr(), n2.data_ptr(), nh.data_ptr())
    torch.cuda.synchronize(); dt = time.time() - t
print("sketch: %.1f ms -> %.0f genomes/s; nhash mean %.1f" % (dt * 1e3, N / dt, nh.float().mean().item()))
for g in (0, 4321, 9999):
    w_hv, w_n2, w_nh = orc.sketch_genome(orc.synth_genome(g, L))
    assert int(nh[g]) == w_nh and int(n2[g]) == w_n2 and bool((hv[g].cpu().numpy() == w_hv).all()), g
print("parity with the CPU oracle on genomes 0, 4321, 9999: ok")
cap = 8_000_000
hits = torch.empty(cap * 3, dtype=torch.int32, device=dev)
for rep in range(3):
    torch.cuda.synchronize(); t = time.time()
    found, st = ctx.dist_dev(hv.data_ptr(), n2.data_ptr(), N, hv.data_ptr(), n2.data_ptr(), N, D, 21, True, 85.0, hits.data_ptr(), cap)
    torch.cuda.synchronize(); dt = time.time() - t
h = hits[: found * 3].view(found, 3)
same = (h[:, 0] // 100 == h[:, 1] // 100)
print("dist (symmetric, th 85): %.2f ms, %d hits, %.1f%% inside a 100-genome cluster, ANI range %.2f..%.2f" % (
    dt * 1e3, found, 100.0 * same.float().mean().item(), h[:, 2].view(torch.float32).min().item(), h[:, 2].view(torch.float32).max().item()))
