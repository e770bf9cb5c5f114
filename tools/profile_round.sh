#!/bin/bash
# One gpurun call for a round's evidence: the whole GPU test suite, every profile (bench command, dist same set / two sets /
# symmetric / large sketches, the realistic legs) and the bench line.  Usage (on the GPU box): tools/profile_round.sh <tag>
# Then, here: tools/derive_prof.py <tag>; tools/kmer_isa.py <tag> [packed]; copy gpurun_out/<tag>/bench_line.json and
# gpurun_out/prof_<tag>real/<tag>_realistic_kernel_stats.txt into profiles/; tools/profiles_readme.py <tag>; commit; and run
# bench.py once more for the line that quotes the committed counters.
TAG=${1:-r05}
cd "${GRAFT_REPO_ROOT:-.}"
mkdir -p gpurun_out/$TAG
timeout 2400 python3 -m pytest tests -m gpu -x -q > gpurun_out/$TAG/pytest_gpu.txt 2>&1; echo "pytest rc=$?" >> gpurun_out/$TAG/pytest_gpu.txt
grep -n "passed\|failed\|rror\|rc=" gpurun_out/$TAG/pytest_gpu.txt | tail -n 4
bash tools/profile_gpu.sh $TAG > gpurun_out/prof_$TAG.log 2>&1
bash tools/profile_dist.sh ${TAG}dist > gpurun_out/prof_${TAG}dist.log 2>&1
HG_DIST_ARGS="--sets two" bash tools/profile_dist.sh ${TAG}dist2 > gpurun_out/prof_${TAG}dist2.log 2>&1
HG_DIST_ARGS="--sets sym" bash tools/profile_dist.sh ${TAG}distsym > gpurun_out/prof_${TAG}distsym.log 2>&1
HG_DIST_ARGS="--nhash 6666" bash tools/profile_dist.sh ${TAG}distf16 > gpurun_out/prof_${TAG}distf16.log 2>&1
bash tools/profile_realistic.sh $TAG > gpurun_out/prof_${TAG}real.log 2>&1
cat gpurun_out/prof_${TAG}*/errors.txt 2>/dev/null
timeout 1500 python3 bench.py --steps 20 --warmup 3 > gpurun_out/$TAG/bench_line.json 2> gpurun_out/$TAG/bench_log.txt; echo "bench rc=$?"
tail -n 22 gpurun_out/$TAG/bench_log.txt
