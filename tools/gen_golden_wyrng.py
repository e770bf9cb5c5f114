#!/usr/bin/env python3
"""Writes tests/golden/kat_wyrng.json: consecutive WyRng outputs for several seeds.

Written from the PUBLISHED definition of wyhash (final version 1) `wyrng`, in Python integers, independently of
oracle/hg_oracle.c and of the HIP kernels (which are the things the vectors check):

    wyrng(seed*):  *seed += 0xa0761d6478bd642f;  return wymum(*seed ^ 0xe7037ed1a0b428db, *seed)
    wymum(a, b):   the 128-bit product a*b, high half XOR low half

The Rust crate the reference uses (wyhash 0.5.0: src/hd.rs:24,44,51,100,103) wraps exactly this: `WyRng(u64)`,
`seed_from_u64(s)` stores s unchanged (no seed expansion), `next_u64` is wyrng on the state.  The crate's README
known answer -- WyRng::seed_from_u64(3).next_u64() == 0x3e99a772750dcbe -- is asserted below before anything is
written, so the generator itself is pinned by the one published vector.
"""
import json
import os

M64 = (1 << 64) - 1
P0 = 0xA0761D6478BD642F
P1 = 0xE7037ED1A0B428DB


def wymum(a, b):
    p = a * b
    return ((p >> 64) ^ p) & M64


class WyRng:
    def __init__(self, seed):
        self.s = seed & M64

    def next_u64(self):
        self.s = (self.s + P0) & M64
        return wymum(self.s ^ P1, self.s)


SEEDS = [3, 0, 1, M64, 0x002BB0CF87D9C549, 0x908794018D1F0246, 123, (M64 - P0 + 1) & M64]  # the last one steps onto state 0
N = 16


def main():
    assert WyRng(3).next_u64() == 0x3E99A772750DCBE, "the published known answer does not hold"
    out = {"definition": "wyhash final v1 wyrng (wyhash crate 0.5.0 WyRng), tools/gen_golden_wyrng.py",
           "seed": 3, "first": "%016x" % WyRng(3).next_u64(), "streams": []}
    for s in SEEDS:
        r = WyRng(s)
        out["streams"].append({"seed": "%016x" % s, "next_u64": ["%016x" % r.next_u64() for _ in range(N)]})
    path = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "tests", "golden", "kat_wyrng.json")
    json.dump(out, open(path, "w"), indent=1)
    print("wrote", os.path.abspath(path))


if __name__ == "__main__":
    main()
