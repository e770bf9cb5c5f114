#!/usr/bin/env python3
"""per-call latency / throughput of hg_kmer_hash_sample vs number of calling threads and link form (debug key "hostfed":
ascii = bases cross the link as ASCII, packed = 2-bit packed by the calling thread, auto = the library's choice).  The
forms are run in turn, pass after pass, so that drift of the shared host hits them alike; median and best of the passes."""
import os, sys, time, threading
import numpy as np, torch
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import hypergen_amd as hg
L = 5_000_000
HF = 128
PASSES = int(os.environ.get("HG_PROBE_PASSES", "7"))
BIND = os.environ.get("HG_PROBE_BIND", "0") == "1"  # the calling threads bound to the device's NUMA node
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
MODES = ("ascii", "packed", "")
for PT in (8, 16, 32) if BIND else (1, 4, 8, 16, 32):
    cs = {m: [hg.Context(0) for _ in range(PT)] for m in MODES}
    outs = [np.zeros(8192, np.uint64) for _ in range(PT)]
    for m in MODES:
        for c in cs[m]:
            if m:
                c.set_debug("hostfed", m)
    res = {m: [] for m in MODES}
    packed_calls = {m: 0 for m in MODES}
    for rep in range(PASSES + 1):
        for m in MODES:
            def w(t):
                n = hg.C.c_size_t(0)
                c = cs[m][t]
                if BIND:
                    hg.lib().hg_bind_thread_to_numa_node(hg.lib().hg_device_numa_node(0), PT)
                for g in range(t, HF, PT):
                    c._ck(hg.lib().hg_kmer_hash_sample(c._h, hg._ptr(rows[g]), rows[g].size, 21, hg.C.c_uint64(thr), hg.C.c_uint64(123), 1, 0,
                                                       hg._ptr(outs[t]), 8192, hg.C.byref(n)))
            ths = [threading.Thread(target=w, args=(t,)) for t in range(PT)]
            t0 = time.perf_counter()
            [x.start() for x in ths]
            [x.join() for x in ths]
            dt = time.perf_counter() - t0
            if rep:
                res[m].append(dt)
                packed_calls[m] += sum(c.last_kernel("kmer").endswith("true>") for c in cs[m])
    for m in MODES:
        r = sorted(res[m])
        print("threads %2d mode %-7s: median %7.0f genomes/s, best %7.0f, worst %7.0f; last calls packed %3.0f %%" % (
            PT, m or "auto", HF / r[len(r) // 2], HF / r[0], HF / r[-1], 100.0 * packed_calls[m] / (PASSES * PT)), flush=True)
    for m in MODES:
        for c in cs[m]:
            c.close()
