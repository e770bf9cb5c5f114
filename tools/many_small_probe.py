#!/usr/bin/env python3
"""Sanity / rate probe: very many tiny genomes in one batch (development aid)."""
import sys, time
sys.path.insert(0, ".")
import torch, numpy as np
import hypergen_amd as hg
from oracle import oracle as orc
dev = torch.device("cuda:0")
ctx = hg.Context(0)
ctx.set_stream(torch.cuda.current_stream().cuda_stream)
p = hg.default_params(scaled=100)
for n, L in ((100_000, 2_000), (20_000, 50_000)):
    stride = (L + 1 + 15) // 16 * 16
    seq = torch.empty(n * stride + 64, dtype=torch.uint8, device=dev)
    ctx.synth_genomes_dev(0, n, L, stride, seq.data_ptr())
    offs = np.arange(n, dtype=np.uint64) * stride
    lens = np.full(n, L + 1, np.uint64)
    hv = torch.empty((n, 4096), dtype=torch.int16, device=dev)
    n2 = torch.empty(n, dtype=torch.int32, device=dev)
    nh = torch.empty(n, dtype=torch.int32, device=dev)
    ctx.enable_timing(True)
    for rep in range(3):
        ctx.timings()
        torch.cuda.synchronize(); t = time.time()
        ctx.sketch_batch_dev(seq.data_ptr(), offs, lens, p, hv.data_ptr(), n2.data_ptr(), nh.data_ptr())
        torch.cuda.synchronize(); dt = time.time() - t
        tm = ctx.timings()
    ok = True
    for g in (0, n // 2, n - 1):
        w_hv, w_n2, w_nh = orc.sketch_genome(orc.synth_genome(g, L), scaled=100)
        ok &= int(nh[g]) == w_nh and int(n2[g]) == w_n2 and bool((hv[g].cpu().numpy() == w_hv).all())
    print("%d x %d bp: %.1f ms -> %.0f genomes/s; kernels %s; parity %s" % (
        n, L, dt * 1e3, n / dt, {k: round(v[0], 2) for k, v in tm.items() if v[1]}, ok))
