#!/usr/bin/env python3
"""End-to-end CLI rate: `hyper-gen sketch` over N synthetic 5 Mbp FASTA files (80-column, page cache), then
`hyper-gen dist` of the sketch file against itself.  Development aid; needs a GPU."""
import argparse
import os
import shutil
import subprocess
import sys
import tempfile
import time

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import torch  # noqa: F401,E402  (HIP runtime load order)
from oracle import oracle as orc  # noqa: E402  (input generation only)

ap = argparse.ArgumentParser()
ap.add_argument("--files", type=int, default=300)
ap.add_argument("--L", type=int, default=5_000_000)
ap.add_argument("--threads", type=int, default=16)
ap.add_argument("--device", default="cpu", help="-D of the CLI: cpu = needletail read semantics, gpu = read_merge_seq")
ap.add_argument("--reps", type=int, default=3)
ap.add_argument("--no-dist", action="store_true")
ap.add_argument("--debug", action="store_true", help="RUST_LOG=debug on the last repetition (stage timings)")
a = ap.parse_args()
root = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..")
exe = os.path.join(root, "hyper-gen_amd", "hyper-gen")
d = tempfile.mkdtemp(prefix="hgcli_", dir="/tmp")
try:
    g = orc.synth_genomes_mt(0, a.files, a.L, a.threads)
    t0 = time.time()
    for i in range(a.files):
        b = g[i][1:].tobytes()
        with open(os.path.join(d, "g%05d.fna" % i), "wb") as f:
            f.write(b">g%d\n" % i)
            f.write(b"\n".join(b[j:j + 80] for j in range(0, len(b), 80)))
            f.write(b"\n")
    print("wrote %d files in %.1f s" % (a.files, time.time() - t0))
    sk = os.path.join(d, "out.sketch")
    for rep in range(a.reps):
        t0 = time.time()
        dbg = a.debug and rep == a.reps - 1
        out = subprocess.run([exe, "sketch", "-p", d, "-o", sk, "-t", str(a.threads), "-D", a.device],
                             stdout=subprocess.PIPE, stderr=None if rep else subprocess.DEVNULL, check=True,
                             env=dict(os.environ, RUST_LOG="debug") if dbg else None).stdout.decode()
        dt = time.time() - t0
        print("\n".join(l for l in out.splitlines() if dbg or "Speed" in l))
        print("hyper-gen sketch: %d files in %.2f s -> %.0f files/s (%.2f GB/s of FASTA)" % (
            a.files, dt, a.files / dt, a.files * a.L * 81 / 80 / dt / 1e9))
    if a.no_dist:
        raise SystemExit(0)
    t0 = time.time()
    subprocess.check_call([exe, "dist", "-r", sk, "-q", sk, "-o", os.path.join(d, "ani.tsv"), "-t", str(a.threads)],
                          stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL)
    print("hyper-gen dist (%d x %d): %.2f s" % (a.files, a.files, time.time() - t0))
finally:
    shutil.rmtree(d, ignore_errors=True)
