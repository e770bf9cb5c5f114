#!/usr/bin/env python3
"""End-to-end `hyper-gen dist` on a synthetic .sketch file of N clustered sketches (development aid; GPU)."""
import argparse
import os
import shutil
import subprocess
import sys
import tempfile
import time

import numpy as np
import torch

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import hypergen_amd as hg  # noqa: E402
import bench  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--n", type=int, default=10000)
ap.add_argument("--threads", type=int, default=16)
a = ap.parse_args()
exe = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "hyper-gen_amd", "hyper-gen")
d = tempfile.mkdtemp(prefix="hgdist_", dir="/tmp")
try:
    hv = bench.clustered_hvs(a.n, 0, torch.device("cuda:0")).cpu().numpy()
    n2 = (hv.astype(np.int64) ** 2).sum(1).astype(np.int32)
    recs = []
    t0 = time.time()
    for i in range(a.n):
        q, packed = hg.hv_pack(hv[i])
        recs.append(dict(ksize=21, scaled=1500, canonical=True, seed=123, hv_d=4096, hv_quant_bits=q,
                         hv_norm_2=int(n2[i]), file_str="/data/genomes/cluster%04d/genome_%06d.fna" % (i // 100, i),
                         hv=packed.view(np.int16)))
    sk = os.path.join(d, "db.sketch")
    hg.write_sketch_file(sk, recs)
    print("wrote %d sketches (%.1f MB) in %.1f s" % (a.n, os.path.getsize(sk) / 1e6, time.time() - t0))
    for rep in range(2):
        out = os.path.join(d, "ani.tsv")
        t0 = time.time()
        o = subprocess.run([exe, "dist", "-r", sk, "-q", sk, "-o", out, "-t", str(a.threads)], check=True,
                           stdout=subprocess.PIPE, stderr=None if rep else subprocess.DEVNULL,
                           env=dict(os.environ, RUST_LOG="debug")).stdout.decode()
        dt = time.time() - t0
        if rep:
            print("\n".join(l.split(" - ", 1)[1] for l in o.splitlines() if " - " in l))
        lines = sum(1 for _ in open(out))
        print("hyper-gen dist %d x %d: %.2f s, %d TSV lines (%.1f MB)" % (a.n, a.n, dt, lines, os.path.getsize(out) / 1e6))
finally:
    shutil.rmtree(d, ignore_errors=True)
