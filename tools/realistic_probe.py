#!/usr/bin/env python3
"""The two `realistic` legs of bench.py on their own (profiling target: `rocprofv3 --kernel-trace --stats -- python3
tools/realistic_probe.py --leg draft|small|clean`): which kernels a draft assembly / a batch of many small genomes spends
its time in, against the clean genomes of the headline.  No oracle, no gates (bench.py has them)."""
import argparse
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import hypergen_amd as hg  # noqa: E402
import bench  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--leg", default="draft", choices=("clean", "draft", "small"))
ap.add_argument("--form", default="packed", choices=("packed", "ascii"))
ap.add_argument("--reps", type=int, default=10)
ap.add_argument("--small-n", type=int, default=100_000)
ap.add_argument("--small-len", type=int, default=50_000)
ap.add_argument("--scaled", type=int, default=1500, help="FracMinHash sampling rate 1/scaled (denser sketches: more hits per work item)")
ap.add_argument("--ksize", type=int, default=21)
ap.add_argument("--hv-d", type=int, default=4096)
ap.add_argument("--non-canonical", action="store_true")
ap.add_argument("--repeat", type=int, default=50_000, help="length of the draft leg's tandem repeat (171-base unit)")
a = ap.parse_args()
dev = torch.device("cuda:0")
ctx = hg.Context(0)
ctx.set_stream(torch.cuda.current_stream().cuda_stream)
N, L = (a.small_n, a.small_len) if a.leg == "small" else (1000, bench.L_GENOME)
stride = (L + 1 + 15) // 16 * 16
seq = torch.empty(N * stride + 64, dtype=torch.uint8, device=dev)
ctx.synth_genomes_dev(0, N, L, stride, seq.data_ptr())
if a.leg == "draft":
    bench.draftify(seq, N, stride, L, repeat_len=a.repeat)
offs, lens = np.arange(N, dtype=np.uint64) * stride, np.full(N, L + 1, np.uint64)
bsz = hg.lib().hg_pack2_size(L + 1)
boffs = np.arange(N, dtype=np.uint64) * bsz
blobs = torch.empty(N * bsz + 64, dtype=torch.uint8, device=dev)
ctx.pack2_batch_dev(seq.data_ptr(), offs, lens, blobs.data_ptr(), boffs)
p = hg.default_params(scaled=a.scaled, ksize=a.ksize, hv_d=a.hv_d, canonical=0 if a.non_canonical else 1)
hv = torch.empty((N, a.hv_d), dtype=torch.int16, device=dev)
n2, nh = torch.empty(N, dtype=torch.int32, device=dev), torch.empty(N, dtype=torch.int32, device=dev)


def step():
    if a.form == "packed":
        ctx.sketch_batch_dev_packed(blobs.data_ptr(), boffs, lens, p, hv.data_ptr(), n2.data_ptr(), nh.data_ptr())
    else:
        ctx.sketch_batch_dev(seq.data_ptr(), offs, lens, p, hv.data_ptr(), n2.data_ptr(), nh.data_ptr())


for _ in range(5):
    step()
ctx.enable_timing(True)
ctx.timings()
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(a.reps):
    step()
torch.cuda.synchronize()
dt = time.perf_counter() - t0
tm = ctx.timings()
print("%s, %s: %d genomes of %d bp, %.1f genomes/s, %.0f Mbase/s; kernel ms per step %s; mean hashes %.1f" % (
    a.leg, a.form, N, L, N * a.reps / dt, N * a.reps * L / dt / 1e6, {k: round(v[0] / a.reps, 3) for k, v in tm.items() if v[1]},
    nh.float().mean().item()))
