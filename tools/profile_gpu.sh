#!/bin/bash
# Run on the GPU box (via gpurun).  Kernel-trace stats of the bench command + separate PMC passes
# (FETCH_SIZE and WRITE_SIZE cannot share a pass on gfx950; --pmc runs carry only --kernel-trace).
# Usage: tools/profile_gpu.sh <round-tag>      outputs under gpurun_out/prof_<tag>/
set -u
TAG=${1:-r01}
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
OUT=$ROOT/gpurun_out/prof_$TAG
rm -rf "$OUT"; mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
# only the headline shapes are launched (1 000-genome sketch, 10k x 10k dist, 50k x 10k search): the 10k-genome leg, the
# host-fed leg and the parity gate would add launches of other sizes to the per-kernel averages
BENCH="python3 $ROOT/bench.py --steps 3 --warmup 1 --no-cpu-baseline --genomes-10k 0 --hostfed-genomes 0 --no-realistic --no-dist-variants --no-cli"
export HG_PROFILE_COMMAND="$BENCH"
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/stats" -- $BENCH > "$OUT/stats.json" 2> "$OUT/stats.log"
for pass in "FETCH_SIZE" "WRITE_SIZE" "SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_SALU SQ_WAVES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY" "GRBM_GUI_ACTIVE SQ_INSTS_VALU_MFMA_MOPS_F16 SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_MFMA SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_ANY"; do
  name=$(echo "$pass" | cut -d' ' -f1)
  timeout 900 rocprofv3 --pmc $pass --kernel-trace --output-format csv -d "$OUT/pmc_$name" -- $BENCH > "$OUT/pmc_$name.json" 2> "$OUT/pmc_$name.log" || echo "pass $name failed" >> "$OUT/errors.txt"
done
cd "$ROOT" && python3 tools/summarize_prof.py "$OUT" "$TAG" > "$OUT/summary.txt" 2>&1
cp "$(find "$OUT/stats" -name '*kernel_stats.csv' | head -1)" "$OUT/${TAG}_rocprofv3_kernel_stats_full.csv"
rm -rf "$OUT"/pmc_*/ "$OUT/stats"   # raw traces stay on the box; the summaries travel back
cat "$OUT/summary.txt"
