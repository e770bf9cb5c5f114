#!/usr/bin/env python3
"""where the one-call-per-genome pattern spends its time: host packing alone, and resident single-genome calls alone, vs threads"""
import os, sys, time, threading
import numpy as np, torch
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import hypergen_amd as hg
L = 5_000_000
HF = 128
dev = torch.device("cuda:0")
ctx = hg.Context(0)
stride = (L + 1 + 15) // 16 * 16
seq = torch.empty(HF * stride + 64, dtype=torch.uint8, device=dev)
ctx.synth_genomes_dev(0, HF, L, stride, seq.data_ptr())
host = torch.empty((HF, L + 1), dtype=torch.uint8).pin_memory()
for g in range(HF):
    host[g].copy_(seq[g * stride: g * stride + L + 1])
rows = [host[g].numpy() for g in range(HF)]
bsz = hg.lib().hg_pack2_size(L + 1)
p = hg.default_params()


def run(PT, fn):
    best = 1e9
    for rep in range(3):
        ths = [threading.Thread(target=fn, args=(t,)) for t in range(PT)]
        t0 = time.perf_counter()
        [x.start() for x in ths]
        [x.join() for x in ths]
        dt = time.perf_counter() - t0
        if rep:
            best = min(best, dt)
    return best


for PT in (1, 4, 8, 16, 32):
    pins = [torch.empty(bsz, dtype=torch.uint8).pin_memory() for _ in range(PT)]

    def w_pack(t):
        for g in range(t, HF, PT):
            hg.lib().hg_pack2(hg._ptr(rows[g]), rows[g].size, 0, hg.C.c_void_p(pins[t].data_ptr()))
    b = run(PT, w_pack)
    print("pack only     threads %2d: %7.0f genomes/s, %.3f ms per genome and thread, %.0f GB/s read" % (PT, HF / b, b * PT / HF * 1e3, HF * L / b / 1e9), flush=True)
    cs = [hg.Context(0) for _ in range(PT)]
    hv = [torch.empty((1, 4096), dtype=torch.int16, device=dev) for _ in range(PT)]
    n2 = [torch.empty(1, dtype=torch.int32, device=dev) for _ in range(PT)]
    nh = [torch.empty(1, dtype=torch.int32, device=dev) for _ in range(PT)]

    def w_dev(t):
        for g in range(t, HF, PT):
            cs[t].sketch_batch_dev(seq.data_ptr(), np.array([g * stride], np.uint64), np.array([L + 1], np.uint64), p, hv[t].data_ptr(), n2[t].data_ptr(), nh[t].data_ptr())
            cs[t].sync()
    b = run(PT, w_dev)
    print("resident call threads %2d: %7.0f genomes/s, %.3f ms per call and thread" % (PT, HF / b, b * PT / HF * 1e3), flush=True)

    def w_up(t):
        st = torch.cuda.Stream()
        d = torch.empty(bsz, dtype=torch.uint8, device=dev)
        with torch.cuda.stream(st):
            for g in range(t, HF, PT):
                d.copy_(pins[t], non_blocking=True)
                st.synchronize()
    b = run(PT, w_up)
    print("upload only   threads %2d: %7.0f blobs/s, %.3f ms per upload and thread, %.1f GB/s" % (PT, HF / b, b * PT / HF * 1e3, HF * bsz / b / 1e9), flush=True)
    for c in cs:
        c.close()
