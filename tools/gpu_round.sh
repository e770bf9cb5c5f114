#!/bin/bash
# One gpurun call: GPU tests, k-mer A/B (development library), bench line.  Outputs under gpurun_out/.
cd "${GRAFT_REPO_ROOT:-.}"
mkdir -p gpurun_out
timeout 1500 python -m pytest tests -m gpu -x -q > gpurun_out/pytest_gpu.txt 2>&1; echo "pytest rc=$?" >> gpurun_out/pytest_gpu.txt
tail -5 gpurun_out/pytest_gpu.txt
if [ -f tools/_exp_libhypergen.so ]; then
  HYPERGEN_LIB=$PWD/tools/_exp_libhypergen.so timeout 300 python tools/quick_bench.py --genomes 1000 --reps 5 --dist 0 --variants 0,4 > gpurun_out/kmer_ab.txt 2>&1
  tail -4 gpurun_out/kmer_ab.txt
fi
timeout 900 python bench.py --steps 20 --warmup 3 > gpurun_out/bench_line.json 2> gpurun_out/bench_log.txt; echo "bench rc=$?"
tail -12 gpurun_out/bench_log.txt
head -c 3000 gpurun_out/bench_line.json
