#!/usr/bin/env python3
"""Kernel time of the k-mer sampling kernel on N resident synthetic genomes (development aid for A/B runs: the library is
chosen with HYPERGEN_LIB, one process per variant, processes alternated on one box).  Prints one line."""
import argparse
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import hypergen_amd as hg  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--genomes", type=int, default=1000)
ap.add_argument("--L", type=int, default=5_000_000)
ap.add_argument("--k", type=int, default=21)
ap.add_argument("--canonical", type=int, default=1)
ap.add_argument("--reps", type=int, default=15)
ap.add_argument("--settle", type=int, default=25)
ap.add_argument("--packed", type=int, default=0, help="1: the genomes are resident as hg_pack2 blobs (hg_sketch_batch_dev_packed)")
a = ap.parse_args()
dev = torch.device("cuda:0")
ctx = hg.Context(0)
ctx.set_stream(torch.cuda.current_stream().cuda_stream)
stride = (a.L + 1 + 15) // 16 * 16
seq = torch.empty(a.genomes * stride + 64, dtype=torch.uint8, device=dev)
ctx.synth_genomes_dev(0, a.genomes, a.L, stride, seq.data_ptr())
offs = np.arange(a.genomes, dtype=np.uint64) * stride
lens = np.full(a.genomes, a.L + 1, np.uint64)
p = hg.default_params(ksize=a.k, canonical=a.canonical)
hv = torch.empty((a.genomes, p.hv_d), dtype=torch.int16, device=dev)
n2 = torch.empty(a.genomes, dtype=torch.int32, device=dev)
nh = torch.empty(a.genomes, dtype=torch.int32, device=dev)
if a.packed:
    bsz = hg.lib().hg_pack2_size(a.L + 1)
    boffs = np.arange(a.genomes, dtype=np.uint64) * bsz
    blobs = torch.empty(a.genomes * bsz + 64, dtype=torch.uint8, device=dev)
    ctx.pack2_batch_dev(seq.data_ptr(), offs, lens, blobs.data_ptr(), boffs)
    del seq
    torch.cuda.empty_cache()

    def step():
        ctx.sketch_batch_dev_packed(blobs.data_ptr(), boffs, lens, p, hv.data_ptr(), n2.data_ptr(), nh.data_ptr())
else:
    def step():
        ctx.sketch_batch_dev(seq.data_ptr(), offs, lens, p, hv.data_ptr(), n2.data_ptr(), nh.data_ptr())
for _ in range(a.settle):
    step()
torch.cuda.synchronize()
ctx.enable_timing(True)
ts = []
for _ in range(a.reps):
    ctx.timings()
    step()
    torch.cuda.synchronize()
    ts.append(ctx.timings()["kmer"][0])
ts.sort()
chk = int(hv.view(torch.int16).to(torch.int64).sum().item()) ^ int(n2.to(torch.int64).sum().item())
print("%-28s k=%d canon=%d: %s median %.3f ms  min %.3f  (%d genomes, nhash mean %.1f, checksum %x)" % (
    os.path.basename(os.environ.get("HYPERGEN_LIB", "libhypergen_hip.so")), a.k, a.canonical, ctx.last_kernel("kmer"),
    ts[len(ts) // 2], ts[0], a.genomes, nh.float().mean().item(), chk & 0xFFFFFFFFFFFF))
