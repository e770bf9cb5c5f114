#!/bin/bash
# PMC passes for the dist kernel only (tools/dist_only.py).  Usage: tools/profile_dist.sh <tag>
set -u
TAG=${1:-dist}
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
OUT=$ROOT/gpurun_out/prof_$TAG
rm -rf "$OUT"; mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
CMD="python3 $ROOT/tools/dist_only.py --reps 4 ${HG_DIST_ARGS:-}"
export HG_PROFILE_COMMAND="$CMD"
# kernel-trace + stats pass first (its own run: --pmc passes carry only --kernel-trace)
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/stats" -- $CMD > "$OUT/stats.log" 2>&1 || echo "stats pass failed" >> "$OUT/errors.txt"
for pass in "FETCH_SIZE" "WRITE_SIZE" "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAIT_INST_LDS SQ_INSTS_VALU SQ_INSTS_LDS" "GRBM_GUI_ACTIVE SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_MFMA SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_LDS SQ_INSTS_SALU SQ_ACTIVE_INST_VALU" "TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum" "SQ_INST_CYCLES_VMEM SQ_INSTS_VMEM SQ_INST_LEVEL_VMEM SQ_INST_LEVEL_LDS SQ_LDS_CMD_FIFO_FULL SQ_LDS_DATA_FIFO_FULL SQ_LDS_ADDR_CONFLICT SQ_WAVES" "TCP_PENDING_STALL_CYCLES_sum TCP_TCC_READ_REQ_sum TCP_TCC_READ_REQ_LATENCY_sum TCP_TOTAL_CACHE_ACCESSES_sum"; do
  name=$(echo "$pass" | cut -d' ' -f1)
  timeout 600 rocprofv3 --pmc $pass --kernel-trace --output-format csv -d "$OUT/pmc_$name" -- $CMD > "$OUT/pmc_$name.log" 2>&1 || echo "pass $name failed" >> "$OUT/errors.txt"
done
cd "$ROOT" && python3 tools/summarize_prof.py "$OUT" "$TAG" > "$OUT/summary.txt" 2>&1
cp "$(find "$OUT/stats" -name '*kernel_stats.csv' | head -1)" "$OUT/${TAG}_rocprofv3_kernel_stats_full.csv" 2>/dev/null
rm -rf "$OUT"/pmc_*/ "$OUT/stats"
grep -E "dist_mfma|prep|decide" "$OUT/summary.txt" 
