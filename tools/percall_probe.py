#!/usr/bin/env python3
"""per-call latency / throughput of hg_kmer_hash_sample vs number of calling threads and link form"""
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
thr = (2**64 - 1) // 1500
for mode in ("ascii", "packed", ""):
    for PT in (1, 2, 4, 8, 16, 32):
        cs = [hg.Context(0) for _ in range(PT)]
        outs = [np.zeros(8192, np.uint64) for _ in range(PT)]
        for c in cs:
            if mode:
                c.set_debug("hostfed", mode)

        def w(t):
            n = hg.C.c_size_t(0)
            for g in range(t, HF, PT):
                cs[t]._ck(hg.lib().hg_kmer_hash_sample(cs[t]._h, hg._ptr(rows[g]), rows[g].size, 21, hg.C.c_uint64(thr), hg.C.c_uint64(123), 1, 0,
                                                        hg._ptr(outs[t]), 8192, hg.C.byref(n)))
        best = 1e9
        for rep in range(3):
            ths = [threading.Thread(target=w, args=(t,)) for t in range(PT)]
            t0 = time.perf_counter()
            [x.start() for x in ths]
            [x.join() for x in ths]
            dt = time.perf_counter() - t0
            if rep:
                best = min(best, dt)
        print("mode %-7s threads %2d: %7.0f genomes/s, %.3f ms per call and thread" % (mode or "auto", PT, HF / best, best * PT / HF * 1e3), flush=True)
        for c in cs:
            c.close()
