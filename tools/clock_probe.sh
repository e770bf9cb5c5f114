#!/bin/bash
# Sustained shader clock of the dist kernel per build variant: GRBM_GUI_ACTIVE (summed over 8 XCDs) / 8 /
# kernel duration.  Usage: tools/clock_probe.sh [lib ...]   (development aid)
set -u
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
OUT=$ROOT/gpurun_out/clock
mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
for lib in "$@"; do
  tag=$(basename "$lib" .so)
  export HYPERGEN_LIB=$lib
  timeout 300 rocprofv3 --pmc GRBM_GUI_ACTIVE --kernel-trace --output-format csv -d "$OUT/$tag" -- python3 $ROOT/tools/dist_only.py --reps 6 --th 101 > "$OUT/$tag.log" 2>&1
  python3 - "$OUT/$tag" "$tag" <<'PY'
import csv, glob, sys
d, tag = sys.argv[1], sys.argv[2]
dur = {}
for f in glob.glob(d + "/**/*kernel_trace.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        if "dist_mfma" in r["Kernel_Name"]:
            dur[r["Dispatch_Id"]] = int(r["End_Timestamp"]) - int(r["Start_Timestamp"])
vals = []
for f in glob.glob(d + "/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        if "dist_mfma" in r["Kernel_Name"] and r["Counter_Name"] == "GRBM_GUI_ACTIVE" and r["Dispatch_Id"] in dur:
            vals.append((float(r["Counter_Value"]) / 8.0, dur[r["Dispatch_Id"]]))
vals = vals[1:]
if vals:
    cyc = sum(v[0] for v in vals) / len(vals); ns = sum(v[1] for v in vals) / len(vals)
    print("%s: %.0f cycles in %.1f us -> %.2f GHz (%d launches)" % (tag, cyc, ns / 1e3, cyc / ns, len(vals)))
else:
    print(tag, "no data")
PY
done
