#!/usr/bin/env python3
"""Run on the GPU box: produces golden hash sets with the REFERENCE's own kernel
(src/cuda_kernel.cu compiled in place by hipcc -> oracle/_ref/) on seeded synthetic inputs.

    gpurun -- python tools/gen_golden_ref_gpu.py        # writes gpurun_out/golden_ref/*.json
then copy the files into tests/golden/ (they are data: inputs are regenerated from the seed,
the expected hash sets are stored as hex).
"""
import json
import os
import subprocess
import sys
import tempfile

import numpy as np

ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..")
sys.path.insert(0, ROOT)
REF = os.path.join(ROOT, "oracle", "_ref")
OUT = os.path.join(ROOT, "gpurun_out", "golden_ref")


def run_ref(seq, k, scaled, seed=123, canonical=True, slots=0):
    with tempfile.TemporaryDirectory() as td:
        fi, fo = os.path.join(td, "s.bin"), os.path.join(td, "o.bin")
        np.ascontiguousarray(seq, np.uint8).tofile(fi)
        # k >= 25: the -O0 build of the same reference source (the optimised builds fault there, oracle/Makefile)
        hsaco = "ref_cuda_kernel.hsaco" if k <= 24 else "ref_cuda_kernel_O0.hsaco"
        subprocess.check_call([os.path.join(REF, "ref_kmer_runner"), os.path.join(REF, hsaco),
                               fi, str(k), str(scaled), str(seed), "1" if canonical else "0", str(slots), fo])
        return np.fromfile(fo, np.uint64)


def main():
    from oracle import oracle as orc
    os.makedirs(OUT, exist_ok=True)
    cases = []
    # (name, genome id, length, k, scaled, canonical, slots) -- slots=0: the reference's own 8
    for name, g, L, k, scaled, canon, slots in [
        ("g2_100k_k21_s1", 1, 100_000, 21, 1, True, 512),
        ("g2_100k_k21_s1500", 1, 100_000, 21, 1500, True, 0),
        ("g2_5m_k21_s1500", 2, 5_000_000, 21, 1500, True, 0),
        ("g2_300k_k21_s100_noncanon", 3, 300_000, 21, 100, False, 32),
        ("g2_200k_k16_s50", 4, 200_000, 16, 50, True, 64),
        ("g2_200k_k24_s50", 5, 200_000, 24, 50, True, 64),
        ("g2_50k_k9_s20", 7, 50_000, 9, 20, True, 128),
        ("g2_50k_k12_s20", 8, 50_000, 12, 20, True, 128),
        # k = 25..32: the reference kernel's `default:` branch = t1ha2's prime_4 stage (-O0 build, see run_ref)
        ("g2_50k_k25_s20", 9, 50_000, 25, 20, True, 128),
        ("g2_50k_k28_s20", 10, 50_000, 28, 20, True, 128),
        ("g2_50k_k31_s20", 11, 50_000, 31, 20, True, 128),
        ("g2_50k_k32_s7_noncanon", 12, 50_000, 32, 7, False, 256),
        ("g2_100k_k29_s1", 13, 100_000, 29, 1, True, 520),
    ]:
        seq = orc.synth_genome(g, L)
        if "100k" in name:  # sprinkle non-bases and lower case
            rng = np.random.default_rng(g)
            seq = seq.copy()
            seq[rng.choice(L, 50, replace=False) + 1] = ord("N")
            seq[5000:6000] = np.char.lower(seq[5000:6000].view("S1")).view(np.uint8)
        try:
            ref = run_ref(seq, k, scaled, canonical=canon, slots=slots)
        except subprocess.CalledProcessError as e:
            print(name, "REFERENCE KERNEL FAILED:", e.returncode)
            continue
        mine = orc.kmer_hash_sample(seq, k, scaled, 123, canon)
        ok = ref.size == mine.size and bool((ref == mine).all())
        print(name, "ref", ref.size, "oracle", mine.size, "equal", ok)
        cases.append({"name": name, "genome": g, "L": L, "k": k, "scaled": scaled, "canonical": canon,
                      "mutated": "100k" in name, "n": int(ref.size), "oracle_equal": ok,
                      "hashes": ["%016x" % int(x) for x in (ref if ref.size <= 4000 else ref[:: max(1, ref.size // 2000)])],
                      "subsampled": bool(ref.size > 4000),
                      "xor": "%016x" % int(np.bitwise_xor.reduce(ref) if ref.size else 0),
                      "sum": "%016x" % (int(ref.astype(object).sum()) % 2**64 if ref.size else 0)})
    json.dump(cases, open(os.path.join(OUT, "g2_ref_kernel.json"), "w"), indent=1)
    print("wrote", OUT)


if __name__ == "__main__":
    main()
