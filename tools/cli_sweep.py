#!/usr/bin/env python3
"""Development aid: one set of FASTA files, `hyper-gen sketch` under several settings (env knobs of a dev build)."""
import os, subprocess, sys, tempfile, time, shutil
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import torch  # noqa
from oracle import oracle as orc
N, L = int(sys.argv[1]), 5_000_000
root = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..")
exe = os.path.join(root, "hyper-gen_amd", "hyper-gen")
d = tempfile.mkdtemp(prefix="hgcli_", dir="/tmp")
try:
    done = 0
    while done < N:
        m = min(800, N - done)
        g = orc.synth_genomes_mt(done, m, L, 32)
        for i in range(m):
            b = g[i][1:].tobytes()
            with open(os.path.join(d, "g%05d.fna" % (done + i)), "wb") as f:
                f.write(b">g%d\n" % i + b"\n".join(b[j:j + 80] for j in range(0, len(b), 80)) + b"\n")
        done += m
    print("wrote", N, flush=True)
    for spec in sys.argv[2:]:
        env = dict(os.environ, RUST_LOG="debug")
        t = "16"
        pre = []
        for kv in spec.split(","):
            k, v = kv.split("=")
            if k == "t":
                t = v
            elif k == "cpus":
                pre = ["taskset", "-c", v]
            else:
                env[k] = v
        for rep in range(2):
            t0 = time.time()
            out = subprocess.run(pre + [exe, "sketch", "-p", d, "-o", os.path.join(d, "o.sketch"), "-t", t], env=env,
                                 stdout=subprocess.PIPE, check=True).stdout.decode()
            dt = time.time() - t0
        print("==", spec, "wall %.2f s" % dt)
        print("\n".join(l.split(" - ", 1)[1] for l in out.splitlines() if "DEBUG" in l or "Speed" in l), flush=True)
finally:
    shutil.rmtree(d, ignore_errors=True)
