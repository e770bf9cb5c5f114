#!/bin/bash
# Issue / wait / LDS counters of the k-mer sampling kernel per build variant (development aid).
# Usage: tools/pmc_kmer.sh <lib.so | prod>...      (run on the GPU box; prints a few lines per library)
set -u
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
OUT=$ROOT/gpurun_out/pmckmer
rm -rf "$OUT"; mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
for lib in "$@"; do
  tag=$(basename "$lib" .so)
  if [ "$lib" = prod ]; then unset HYPERGEN_LIB; else export HYPERGEN_LIB=$lib; fi
  i=0
  for pass in "GRBM_GUI_ACTIVE SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_ACTIVE_INST_ANY SQ_INSTS_LDS SQ_INST_LEVEL_LDS" \
              "SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_INSTS_SALU SQ_ACTIVE_INST_SCA SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_LDS SQ_BUSY_CYCLES" \
              "SQ_INSTS_VMEM SQ_INST_LEVEL_VMEM SQ_LDS_CMD_FIFO_FULL SQ_LDS_DATA_FIFO_FULL SQ_LDS_ADDR_CONFLICT SQ_WAVES SQ_INST_CYCLES_VMEM SQ_THREAD_CYCLES_VALU"; do
    i=$((i+1))
    timeout 300 rocprofv3 --pmc $pass --kernel-trace --output-format csv -d "$OUT/$tag/p$i" -- python3 $ROOT/tools/kmer_time.py --reps 4 --settle 3 ${KARGS:-} > "$OUT/$tag.p$i.log" 2>&1
  done
  python3 - "$OUT/$tag" "$tag" <<'PY'
import csv, glob, sys, collections
d, tag = sys.argv[1], sys.argv[2]
acc = collections.defaultdict(list)
name = None
for f in glob.glob(d + "/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        if "kmer_sample" in r["Kernel_Name"]:
            name = r["Kernel_Name"].split("(")[0]
            acc[r["Counter_Name"]].append(float(r["Counter_Value"]))
v = {k: sum(x[1:]) / max(1, len(x) - 1) for k, x in acc.items()}
cyc = v.get("GRBM_GUI_ACTIVE", 0) / 8
kmers = 1000 * 4_999_981
print(tag, name, "cycles %.0f  = %.1f per wave-k-mer and SIMD" % (cyc, cyc / (kmers / 64 / 1024)))
print("   VALU %.2f / k-mer, %.3f / SIMD / cycle; SALU %.2f / k-mer; LDS %.2f / k-mer; VMEM %.3f / k-mer" % (
    v.get("SQ_INSTS_VALU", 0) * 64 / kmers, v.get("SQ_INSTS_VALU", 0) / max(1, cyc) / 1024, v.get("SQ_INSTS_SALU", 0) * 64 / kmers,
    v.get("SQ_INSTS_LDS", 0) * 64 / kmers, v.get("SQ_INSTS_VMEM", 0) * 64 / kmers))
print("   wave split: issue %.0f%% wait_inst %.0f%% wait_any %.0f%% (wait_lds %.0f%%); waves %.0f" % (tuple(
    100 * v.get(k, 0) / max(1, v.get("SQ_WAVE_CYCLES", 1)) for k in ("SQ_ACTIVE_INST_ANY", "SQ_WAIT_INST_ANY", "SQ_WAIT_ANY", "SQ_WAIT_INST_LDS")) + (v.get("SQ_WAVES", 0),)))
print("   busy: VALU active %.0f%% of SQ_BUSY*4?, LDS active/CU/cycle %.2f, LDS idx %.2f, bank conflict %.2f of idx; LDS avg latency %.0f cyc; fifo full cmd %.3g data %.3g" % (
    100 * v.get("SQ_ACTIVE_INST_VALU", 0) / max(1, v.get("SQ_BUSY_CYCLES", 1)), v.get("SQ_ACTIVE_INST_LDS", 0) / max(1, cyc) / 256,
    v.get("SQ_LDS_IDX_ACTIVE", 0) / max(1, cyc) / 256, v.get("SQ_LDS_BANK_CONFLICT", 0) / max(1, v.get("SQ_LDS_IDX_ACTIVE", 1)),
    4 * v.get("SQ_INST_LEVEL_LDS", 0) / max(1, v.get("SQ_INSTS_LDS", 1)), v.get("SQ_LDS_CMD_FIFO_FULL", 0), v.get("SQ_LDS_DATA_FIFO_FULL", 0)))
print("   raw:", {k: "%.4g" % x for k, x in sorted(v.items())})
PY
  rm -rf "$OUT/$tag"
done
