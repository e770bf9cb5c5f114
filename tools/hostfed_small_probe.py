#!/usr/bin/env python3
"""Host-fed sketch rate for very many small genomes (hg_sketch_batch packs them per sub-batch)."""
import sys, time
sys.path.insert(0, ".")
import numpy as np, torch
import hypergen_amd as hg
rng = np.random.default_rng(1)
n, L = 50_000, 2_000
big = rng.choice(np.frombuffer(b"ACGT", np.uint8), n * L)
gs = [big[i * L:(i + 1) * L] for i in range(n)]
ctx = hg.Context(0)
p = hg.default_params(scaled=100)
ctx.sketch_batch(gs[:100], p)
for rep in range(3):
    t = time.time(); hv, n2, nh = ctx.sketch_batch(gs, p); dt = time.time() - t
    print("host-fed %d x %d bp: %.1f ms -> %.0f genomes/s (includes the Python ctypes marshalling of %d pointers)" % (n, L, dt * 1e3, n / dt, n))
