#!/bin/bash
# LDS / VMEM latency and stall counters of the dist kernel per build variant (development aid).
# Usage: tools/pmc_probe.sh lib...
set -u
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
OUT=$ROOT/gpurun_out/pmcprobe
rm -rf "$OUT"; mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
for lib in "$@"; do
  tag=$(basename "$lib" .so)
  export HYPERGEN_LIB=$lib
  i=0
  for pass in "GRBM_GUI_ACTIVE SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_ACTIVE_INST_ANY SQ_INSTS_LDS SQ_INST_LEVEL_LDS" "SQ_INSTS_VMEM SQ_INST_LEVEL_VMEM SQ_LDS_CMD_FIFO_FULL SQ_LDS_DATA_FIFO_FULL SQ_LDS_ADDR_CONFLICT SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_LDS"; do
    i=$((i+1))
    timeout 300 rocprofv3 --pmc $pass --kernel-trace --output-format csv -d "$OUT/$tag/p$i" -- python3 $ROOT/tools/dist_only.py --reps 4 --th 101 > "$OUT/$tag.p$i.log" 2>&1
  done
  python3 - "$OUT/$tag" "$tag" <<'PY'
import csv, glob, sys, collections
d, tag = sys.argv[1], sys.argv[2]
acc = collections.defaultdict(list)
for f in glob.glob(d + "/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        if "dist_mfma" in r["Kernel_Name"]:
            acc[r["Counter_Name"]].append(float(r["Counter_Value"]))
v = {k: sum(x[1:]) / max(1, len(x) - 1) for k, x in acc.items()}
cyc = v.get("GRBM_GUI_ACTIVE", 0) / 8
print(tag, "cycles %.0f" % cyc)
print("   wave split: issue %.0f%% wait_inst %.0f%% wait_any %.0f%% wait_lds %.0f%%" % tuple(
    100 * v.get(k, 0) / max(1, v.get("SQ_WAVE_CYCLES", 1)) for k in ("SQ_ACTIVE_INST_ANY", "SQ_WAIT_INST_ANY", "SQ_WAIT_ANY", "SQ_WAIT_INST_LDS")))
print("   LDS: insts %.3g  avg latency %.0f cyc  idx_active/cycle/CU %.2f  cmd_fifo_full %.3g data_fifo_full %.3g addr_conf %.3g bank_conf %.3g" % (
    v.get("SQ_INSTS_LDS", 0), 4 * v.get("SQ_INST_LEVEL_LDS", 0) / max(1, v.get("SQ_INSTS_LDS", 1)),
    v.get("SQ_LDS_IDX_ACTIVE", 0) / max(1, cyc) / 256, v.get("SQ_LDS_CMD_FIFO_FULL", 0), v.get("SQ_LDS_DATA_FIFO_FULL", 0),
    v.get("SQ_LDS_ADDR_CONFLICT", 0), v.get("SQ_LDS_BANK_CONFLICT", 0)))
print("   VMEM: insts %.3g  avg latency %.0f cyc" % (v.get("SQ_INSTS_VMEM", 0), 4 * v.get("SQ_INST_LEVEL_VMEM", 0) / max(1, v.get("SQ_INSTS_VMEM", 1))))
PY
  rm -rf "$OUT/$tag"
done
