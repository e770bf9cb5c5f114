#!/bin/bash
# rocprofv3 --kernel-trace --stats of the `realistic` legs (tools/realistic_probe.py), one run per leg; per-kernel averages
# into gpurun_out/prof_<tag>real/<tag>_realistic_kernel_stats.txt.   Usage: tools/profile_realistic.sh <tag>
set -u
TAG=${1:-r05}
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
OUT=$ROOT/gpurun_out/prof_${TAG}real
rm -rf "$OUT"; mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
for leg in clean draft small; do
  for form in packed ascii; do
    timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/$leg$form" -- python3 $ROOT/tools/realistic_probe.py --leg $leg --form $form > "$OUT/$leg$form.log" 2>&1
    {
      echo "== $leg genomes, $form-resident: $(grep genomes/s "$OUT/$leg$form.log")"
      python3 - "$OUT/$leg$form" <<'PY'
import csv, glob, sys
rows = []
for f in glob.glob(sys.argv[1] + "/**/*kernel_stats.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        name = r["Name"].replace("void ", "").replace("(anonymous namespace)::", "")
        rows.append((int(r["TotalDurationNs"]), name.split("(")[0][:90], int(r["Calls"]), float(r["AverageNs"])))
tot = sum(r[0] for r in rows) or 1
for t, name, calls, avg in sorted(rows, reverse=True)[:8]:
    print("   %5.1f %%  %-90s calls %4d  avg %10.1f us" % (100.0 * t / tot, name, calls, avg / 1e3))
PY
    } >> "$OUT/${TAG}_realistic_kernel_stats.txt"
    rm -rf "$OUT/$leg$form"
  done
done
cat "$OUT/${TAG}_realistic_kernel_stats.txt"
