#!/bin/bash
# rocprofv3 --kernel-trace --stats of `hyper-gen dist -r A -q B`, `hyper-gen search` (two synthetic 10 000-sketch files) and `hyper-gen sketch`:
# which kernels an end-to-end comparison launches and how long each runs (the new ones of round 5: hg_hv_unpack_kernel, the
# radix sort).  Usage (GPU box): tools/profile_cli.sh <tag>   ->  gpurun_out/prof_<tag>cli/<tag>_cli_kernel_stats.txt
set -u
TAG=${1:-r05}
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
OUT=$ROOT/gpurun_out/prof_${TAG}cli
rm -rf "$OUT"; mkdir -p "$OUT"
D=$(mktemp -d /tmp/hgcli_XXXX)
python3 - "$ROOT" "$D" <<'PY'
import sys, os, numpy as np, torch
sys.path.insert(0, sys.argv[1]); sys.path.insert(0, os.path.join(sys.argv[1], "tools"))
import cli_dist_bench as cb, bench
dev = torch.device("cuda:0")
cb.write_db(os.path.join(sys.argv[2], "a.sketch"), bench.clustered_hvs(10000, 0, dev).cpu().numpy(), 0)
cb.write_db(os.path.join(sys.argv[2], "b.sketch"), bench.clustered_hvs(10000, 0, dev, salt=1).cpu().numpy(), 10000)
PY
cd /tmp && export TMPDIR=/tmp
for what in dist search; do
  extra=""; [ $what = search ] && extra="-n 5"
  timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/$what" -- $ROOT/hyper-gen_amd/hyper-gen $what -r $D/a.sketch -q $D/b.sketch -o $D/out.tsv -t 16 $extra > "$OUT/$what.log" 2>&1
  {
    echo "== hyper-gen $what -r a.sketch -q b.sketch $extra   (10 000 x 10 000 sketches; $(grep -o 'took [0-9.]*s' "$OUT/$what.log" | tail -n 1) under the profiler)"
    python3 - "$OUT/$what" <<'PY'
import csv, glob, sys
rows = []
for f in glob.glob(sys.argv[1] + "/**/*kernel_stats.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        name = r["Name"].replace("void ", "").replace("(anonymous namespace)::", "")
        rows.append((int(r["TotalDurationNs"]), name.split("(")[0][:84], int(r["Calls"]), float(r["AverageNs"])))
tot = sum(r[0] for r in rows) or 1
print("   all kernels together: %.3f ms" % (tot / 1e6))
for t, name, calls, avg in sorted(rows, reverse=True)[:14]:
    print("   %5.1f %%  %-84s calls %3d  avg %9.1f us" % (100.0 * t / tot, name, calls, avg / 1e3))
PY
  } >> "$OUT/${TAG}_cli_kernel_stats.txt"
  rm -rf "$OUT/$what"
done
# `hyper-gen sketch` over 2 048 FASTA files of 5 Mbp (64 distinct genomes + links, as in tools/cli_dist_bench.py): how much of the
# run's wall time the device is busy at all -- the tool is bound by its 16 reader threads and the HIP bring-up
python3 - "$ROOT" "$D" <<'PY'
import sys, os, numpy as np, torch
sys.path.insert(0, sys.argv[1])
import hypergen_amd as hg
L, nd, nf = 5_000_000, 64, 2048
fdir = os.path.join(sys.argv[2], "fasta")
os.mkdir(fdir)
dev = torch.device("cuda:0")
with hg.Context(0) as ctx:
    stride = (L + 1 + 15) // 16 * 16
    seq = torch.empty(nd * stride + 64, dtype=torch.uint8, device=dev)
    ctx.synth_genomes_dev(0, nd, L, stride, seq.data_ptr())
    torch.cuda.synchronize()
    host = seq[: nd * stride].view(nd, stride).cpu().numpy()
nl = np.full((L // 80, 1), 10, np.uint8)
for i in range(nd):
    with open(os.path.join(fdir, "g%05d.fna" % i), "wb") as f:
        f.write(b">g%d\n" % i)
        np.concatenate([host[i, 1:L + 1].reshape(L // 80, 80), nl], axis=1).tofile(f)
for i in range(nd, nf):
    os.symlink("g%05d.fna" % (i % nd), os.path.join(fdir, "g%05d.fna" % i))
PY
cd /tmp
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/sketch" -- $ROOT/hyper-gen_amd/hyper-gen sketch -p $D/fasta -o $D/out.sketch -t 16 > "$OUT/sketch.log" 2>&1
{
  echo "== hyper-gen sketch -p DIR   (2 048 FASTA files of 5 Mbp; $(grep -o 'took [0-9.]*s - Speed: [0-9.]* files/s' "$OUT/sketch.log" | tail -n 1) under the profiler)"
  python3 - "$OUT/sketch" <<'PY'
import csv, glob, sys
rows = []
for f in glob.glob(sys.argv[1] + "/**/*kernel_stats.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        name = r["Name"].replace("void ", "").replace("(anonymous namespace)::", "")
        rows.append((int(r["TotalDurationNs"]), name.split("(")[0][:84], int(r["Calls"]), float(r["AverageNs"])))
tot = sum(r[0] for r in rows) or 1
print("   all kernels together: %.3f ms" % (tot / 1e6))
for t, name, calls, avg in sorted(rows, reverse=True)[:10]:
    print("   %5.1f %%  %-84s calls %3d  avg %9.1f us" % (100.0 * t / tot, name, calls, avg / 1e3))
PY
} >> "$OUT/${TAG}_cli_kernel_stats.txt"
rm -rf "$OUT/sketch"
rm -rf "$D"
cat "$OUT/${TAG}_cli_kernel_stats.txt"
