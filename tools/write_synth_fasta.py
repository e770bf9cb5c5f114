#!/usr/bin/env python3
"""write_synth_fasta.py OUT.fna GENOME_INDEX LENGTH -- one synthetic genome of this repository (SURVEY.md 8d generator,
oracle.synth_genome) as an 80-column FASTA with the header `>g<index>`.  Used by tools/make_ref_sketch.md."""
import os
import sys

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from oracle import oracle as orc  # noqa: E402


def main():
    out, g, L = sys.argv[1], int(sys.argv[2]), int(sys.argv[3])
    s = bytes(orc.synth_genome(g, L)[1:]).decode()
    with open(out, "w") as f:
        f.write(">g%d\n" % g)
        for i in range(0, len(s), 80):
            f.write(s[i:i + 80] + "\n")


if __name__ == "__main__":
    main()
