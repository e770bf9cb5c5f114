#!/bin/bash
# How small a genome can get before the per-genome costs show: tools/realistic_probe.py --leg small over a ladder of sizes
# (packed-resident; HIP-event kernel times per step).  Usage (GPU box): tools/small_genome_sweep.sh > gpurun_out/small_sweep.txt
cd "${GRAFT_REPO_ROOT:-.}"
for spec in "400000 2000" "400000 5000" "200000 10000" "100000 20000" "100000 50000" "20000 250000" "1000 5000000"; do
  set -- $spec
  timeout 300 python3 tools/realistic_probe.py --leg small --small-n $1 --small-len $2 --reps 5 2>&1 | tail -n 1
done
