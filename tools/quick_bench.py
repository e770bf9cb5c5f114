#!/usr/bin/env python3
"""Quick device-resident timing of the sketch and dist paths (development aid, not bench.py)."""
import argparse
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import hypergen_amd as hg  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--genomes", type=int, default=200)
    ap.add_argument("--L", type=int, default=5_000_000)
    ap.add_argument("--reps", type=int, default=3)
    ap.add_argument("--dist", type=int, default=4096, help="R=Q for the dist timing (0 = skip)")
    ap.add_argument("--k", type=int, default=21)
    a = ap.parse_args()
    dev = torch.device("cuda:0")
    ctx = hg.Context(0)
    ctx.set_stream(torch.cuda.current_stream().cuda_stream)
    stride = (a.L + 1 + 15) // 16 * 16
    seq = torch.empty(a.genomes * stride + 64, dtype=torch.uint8, device=dev)
    ctx.synth_genomes_dev(0, a.genomes, a.L, stride, seq.data_ptr())
    torch.cuda.synchronize()
    offs = np.arange(a.genomes, dtype=np.uint64) * stride
    lens = np.full(a.genomes, a.L + 1, np.uint64)
    p = hg.default_params(ksize=a.k)
    hv = torch.empty((a.genomes, p.hv_d), dtype=torch.int16, device=dev)
    n2 = torch.empty(a.genomes, dtype=torch.int32, device=dev)
    nh = torch.empty(a.genomes, dtype=torch.int32, device=dev)
    for rep in range(a.reps + 1):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        ctx.sketch_batch_dev(seq.data_ptr(), offs, lens, p, hv.data_ptr(), n2.data_ptr(), nh.data_ptr())
        torch.cuda.synchronize()
        dt = time.perf_counter() - t0
        if rep:
            print("sketch: %d genomes x %d bp in %.2f ms -> %.0f genomes/s, %.1f GB/s, nhash mean %.1f" % (
                a.genomes, a.L, dt * 1e3, a.genomes / dt, a.genomes * (a.L + 2 * p.hv_d) / dt / 1e9,
                nh.float().mean().item()))
    if a.dist:
        R = a.dist
        g = torch.Generator(device=dev)
        g.manual_seed(1)
        n = 3333
        hvs = (2 * torch.distributions.Binomial(n, torch.tensor(0.5, device=dev)).sample((R, 4096)) - n).to(torch.int16)
        nn = (hvs.int() ** 2).sum(1).int()
        out = torch.empty((R, R), dtype=torch.float32, device=dev)
        for rep in range(a.reps + 1):
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            ctx.dist_full_dev(hvs.data_ptr(), nn.data_ptr(), R, hvs.data_ptr(), nn.data_ptr(), R, 4096, 21, out.data_ptr())
            torch.cuda.synchronize()
            dt = time.perf_counter() - t0
            if rep:
                print("dist: %dx%d in %.2f ms -> %.1f M pairs/s, %.1f TFLOP/s" % (
                    R, R, dt * 1e3, R * R / dt / 1e6, R * R * 8192 / dt / 1e12))
        print("diag ANI", out.diagonal()[:4].tolist(), "offdiag", out[0, 1].item())


if __name__ == "__main__":
    main()
