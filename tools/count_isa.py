#!/usr/bin/env python3
"""Static instruction histogram of one kernel of a HIP file (development aid).
usage: tools/count_isa.py <file.hip> <mangled-name-substring> [extra hipcc flags...]"""
import collections
import subprocess
import sys
import tempfile

src, pat = sys.argv[1], sys.argv[2]
with tempfile.NamedTemporaryFile(suffix=".s") as f:
    subprocess.check_call(["hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-ffp-contract=off", "-S",
                           "--cuda-device-only", "-o", f.name, src] + sys.argv[3:], stderr=subprocess.DEVNULL)
    lines = open(f.name).read().split("\n")
start = [i for i, l in enumerate(lines) if l.startswith("_Z") and pat in l and l.rstrip().split(":")[0].endswith(l.split(":")[0])][0]
end = start
while "s_endpgm" not in lines[end]:
    end += 1
ops = collections.Counter()
for line in lines[start + 1:end]:
    line = line.strip()
    if not line or line[0] in ";." or line.endswith(":"):
        continue
    ops[line.split()[0]] += 1
half = ("v_mad_u64", "v_mul_lo", "v_mul_hi", "v_perm", "v_alignb", "v_lshl_add_u64", "v_lshrrev_b64", "v_lshlrev_b64",
        "v_cmp_lt_u64", "v_cmp_gt_u64", "v_ashrrev_i64", "v_mul_u32_u24")
valu = sum(c for o, c in ops.items() if o.startswith("v_"))
hr = sum(c for o, c in ops.items() if o.startswith(half))
print("total %d  valu %d (half-rate %d)  salu %d  est cycles %d" % (
    sum(ops.values()), valu, hr, sum(c for o, c in ops.items() if o.startswith("s_")), 2 * (valu - hr) + 4 * hr))
for o, c in ops.most_common(28):
    print("  %-26s %d" % (o, c))
for l in lines[end:end + 400]:
    if any(k in l for k in (".vgpr_count", ".sgpr_count", "scratch", ".private_segment_fixed_size")):
        print(l.strip())
        if ".vgpr_count" in l:
            break
