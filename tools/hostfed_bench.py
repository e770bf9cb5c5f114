#!/usr/bin/env python3
"""PCIe-inclusive sketch rate: host buffers in, host results out (hg_sketch_batch)."""
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import hypergen_amd as hg  # noqa: E402
from oracle import oracle as orc  # noqa: E402  (input generation only)

n, L = 256, 5_000_000
cores = min(os.cpu_count() or 1, 32)
g = orc.synth_genomes_mt(0, n, L, cores)
ctx = hg.Context(0)
p = hg.default_params()
for label, arr in (("pageable", g), ("pinned", torch.from_numpy(g).pin_memory().numpy())):
    seqs = [arr[i] for i in range(n)]
    ctx.sketch_batch(seqs[:8], p)
    for rep in range(3):
        t0 = time.perf_counter()
        hv, n2, nh = ctx.sketch_batch(seqs, p)
        dt = time.perf_counter() - t0
    print("host-fed (%s): %d genomes in %.1f ms -> %.0f genomes/s, %.1f GB/s of sequence" % (
        label, n, dt * 1e3, n / dt, n * (L + 1) / dt / 1e9))
