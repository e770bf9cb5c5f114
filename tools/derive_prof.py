#!/usr/bin/env python3
"""profiles/<tag>_derived.md, <tag>_kmer_traffic.json and <tag>_dist_traffic.json from the PMC summaries that
tools/profile_gpu.sh (bench.py run) and tools/profile_dist.sh (tools/dist_only.py run) leave in gpurun_out/.
usage: tools/derive_prof.py <tag>      (reads gpurun_out/prof_<tag>/ and gpurun_out/prof_<tag>dist/)"""
import json
import os
import shutil
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
tag = sys.argv[1] if len(sys.argv) > 1 else "r01"
src, srcd = os.path.join(ROOT, "gpurun_out", "prof_" + tag), os.path.join(ROOT, "gpurun_out", "prof_" + tag + "dist")
out = os.path.join(ROOT, "profiles")
import subprocess  # noqa: E402
sys.path.insert(0, ROOT)
import hypergen_amd as hg  # noqa: E402
HEAD = subprocess.run(["git", "rev-parse", "HEAD"], cwd=ROOT, capture_output=True, text=True).stdout.strip()
DIRTY = bool(subprocess.run(["git", "status", "--porcelain", "--", "hyper-gen_amd/csrc", "include"], cwd=ROOT, capture_output=True,
                            text=True).stdout.strip())


def stamped(d):
    """complete a summary's `_stamp` (written on the GPU box by tools/summarize_prof.py: source hash + kernel names) with
    the commit it was taken from; refuses a summary whose sources are not the ones in this tree"""
    st = d.get("_stamp") or {}
    if st.get("source_sha") != hg.source_stamp():
        raise SystemExit("profile was taken on other sources (%s) than this tree (%s): re-profile" % (st.get("source_sha"), hg.source_stamp()))
    st.update(head=HEAD, sources_dirty_at_copy=DIRTY)
    d["_stamp"] = st
    return d


pmc = stamped(json.load(open(os.path.join(src, tag + "_pmc.json"))))
pmcd = stamped(json.load(open(os.path.join(srcd, tag + "dist_pmc.json")))) if os.path.exists(srcd) else {}
SIMDS = 1024

for f in (tag + "_kernel_stats.csv", tag + "_rocprofv3_kernel_stats_full.csv"):
    shutil.copy(os.path.join(src, f), os.path.join(out, f))
json.dump(pmc, open(os.path.join(out, tag + "_pmc.json"), "w"), indent=1, sort_keys=True)
if pmcd:
    json.dump(pmcd, open(os.path.join(out, tag + "_dist_pmc.json"), "w"), indent=1, sort_keys=True)
    shutil.copy(os.path.join(srcd, tag + "dist_kernel_stats.csv"), os.path.join(out, tag + "_dist_kernel_stats.csv")) \
        if os.path.exists(os.path.join(srcd, tag + "dist_kernel_stats.csv")) else None
# optional third run: the same dist command on LARGE sketches (HG_DIST_ARGS="--nhash 10000": the centred f16 operand path)
srcf = os.path.join(ROOT, "gpurun_out", "prof_" + tag + "distf16")
pmcf = stamped(json.load(open(os.path.join(srcf, tag + "distf16_pmc.json")))) if os.path.exists(srcf) else {}
if pmcf:
    json.dump(pmcf, open(os.path.join(out, tag + "_distf16_pmc.json"), "w"), indent=1, sort_keys=True)
    shutil.copy(os.path.join(srcf, tag + "distf16_kernel_stats.csv"), os.path.join(out, tag + "_distf16_kernel_stats.csv"))


# round 5: the same dist command with R and Q in different buffers (--sets two) and with symmetric = 1 (--sets sym)
extra = {}
for suffix, what in (("dist2", "two distinct sets"), ("distsym", "symmetric = 1")):
    sx = os.path.join(ROOT, "gpurun_out", "prof_" + tag + suffix)
    if os.path.exists(os.path.join(sx, tag + suffix + "_pmc.json")):
        px = stamped(json.load(open(os.path.join(sx, tag + suffix + "_pmc.json"))))
        json.dump(px, open(os.path.join(out, tag + "_" + suffix + "_pmc.json"), "w"), indent=1, sort_keys=True)
        shutil.copy(os.path.join(sx, tag + suffix + "_kernel_stats.csv"), os.path.join(out, tag + "_" + suffix + "_kernel_stats.csv"))
        extra[what] = px


def row(name, d):
    cyc = d["GRBM_GUI_ACTIVE"] / 8.0
    wave = d.get("SQ_WAVE_CYCLES", 0.0)
    issue = d.get("SQ_ACTIVE_INST_ANY", 0.0) / wave if wave else float("nan")
    wany = d.get("SQ_WAIT_ANY", 0.0) / wave if wave else float("nan")
    winst = d.get("SQ_WAIT_INST_ANY", 0.0) / wave if wave else float("nan")
    lds_act = d.get("SQ_LDS_IDX_ACTIVE", 0.0)
    conf = d.get("SQ_LDS_BANK_CONFLICT", 0.0) / lds_act if lds_act else 0.0
    l2 = ""
    if "TCC_HIT_sum" in d:
        l2 = "%.0f %%" % (100 * d["TCC_HIT_sum"] / max(1.0, d["TCC_HIT_sum"] + d["TCC_MISS_sum"]))
    lds_busy = lds_act / (256.0 * cyc) if cyc else 0.0  # SQ_LDS_IDX_ACTIVE per CU and cycle (the LDS array is busy that share of the time)
    vmem = ""
    if "SQ_INST_CYCLES_VMEM" in d and d.get("SQ_INSTS_VMEM"):
        vmem = "%.0f" % (d["SQ_INST_CYCLES_VMEM"] / d["SQ_INSTS_VMEM"])
    tcp = ""
    if "TCP_PENDING_STALL_CYCLES_sum" in d:
        tcp = "%.2f" % (d["TCP_PENDING_STALL_CYCLES_sum"] / (256.0 * cyc))
    return "| `%s` | %.2f | %.3f | %.2f | %.2f | %.2f | %s | %s | %s | %.0f %% / %.0f %% / %.0f %% | %.3g |" % (
        name, cyc / 1e6, d.get("SQ_INSTS_VALU", 0.0) / (SIMDS * cyc), d.get("SQ_VALU_MFMA_BUSY_CYCLES", 0.0) / (SIMDS * cyc),
        lds_busy, conf, l2 or "-", vmem or "-", tcp or "-", 100 * issue, 100 * winst, 100 * wany, d.get("hbm_bytes_per_launch", float("nan")))


lines = ["# Derived metrics from `%s_pmc.json` / `%s_dist_pmc.json` (per launch)" % (tag, tag), "",
         "Source: separate `rocprofv3 --pmc` passes of `bench.py --steps 3 --warmup 1` (`tools/profile_gpu.sh`) and of",
         "`tools/dist_only.py` (`tools/profile_dist.sh`).  `GRBM_GUI_ACTIVE` is summed over the 8 XCDs (/8 = kernel cycles);",
         "SIMDs = 1024; the wave-time split is SQ_ACTIVE_INST_ANY / SQ_WAIT_INST_ANY / SQ_WAIT_ANY over SQ_WAVE_CYCLES; HBM bytes = FETCH_SIZE*1024*2 (gfx950 wide-read correction)",
         "+ WRITE_SIZE*1024.  Generated by `tools/derive_prof.py`.", "",
         "LDS busy = SQ_LDS_IDX_ACTIVE / (256 CUs x kernel cycles) (the unit is settled by the Hamming kernel of round 2: 672 M / 256 = 2.62 M of 10.86 M cycles",
         "= 171 GB of fragment reads at 256 B/clk); VMEM cycles / instr = SQ_INST_CYCLES_VMEM / SQ_INSTS_VMEM; TCP pending stall = TCP_PENDING_STALL_CYCLES / (256 x cycles).", "",
         "| kernel | cycles (M) | VALU instr / SIMD / cycle | MFMA busy | LDS busy | LDS bank-conflict cycles / LDS active | L2 hit | VMEM cycles / instr | TCP pending stall / CU / cycle | wave time: issuing / waiting to issue / waiting on waitcnt | HBM-side bytes |",
         "|---|---|---|---|---|---|---|---|---|---|---|"]
seen = set()
for name, d in sorted(((k, v) for k, v in pmc.items() if not k.startswith("_")), key=lambda kv: -kv[1].get("GRBM_GUI_ACTIVE", 0)):
    if "GRBM_GUI_ACTIVE" in d and "dist_mfma" not in name and "prep" not in name and "decide" not in name:
        lines.append(row(name, d))
        seen.add(name)
for name, d in sorted(((k, v) for k, v in pmc.items() if not k.startswith("_")), key=lambda kv: -kv[1].get("GRBM_GUI_ACTIVE", 0)):
    # the Hamming search of the bench run (template arguments 7 / 8: HAM / FP4)
    if "GRBM_GUI_ACTIVE" in d and name.startswith("dist_mfma_kernel") and name.rstrip(">").split(",")[6].strip() == "true":
        lines.append(row(name + " (bench run: Hamming search 50 000 x 10 000 x 16384)", d))
for name, d in sorted(((k, v) for k, v in pmcd.items() if not k.startswith("_")), key=lambda kv: -kv[1].get("GRBM_GUI_ACTIVE", 0)):
    if "GRBM_GUI_ACTIVE" in d:
        lines.append(row(name + " (dist_only run)", d))
for name, d in sorted(((k, v) for k, v in pmcf.items() if not k.startswith("_")), key=lambda kv: -kv[1].get("GRBM_GUI_ACTIVE", 0)):
    if "GRBM_GUI_ACTIVE" in d and (d.get("SQ_INSTS_MFMA", 0) > 0 or "prep_cen" in name):
        lines.append(row(name + " (dist_only run, large sketches: %s)" % (pmcf["_stamp"].get("command", "").split("dist_only.py")[-1].strip() or "?"), d))
for what, px in extra.items():
    for name, d in sorted(((k, v) for k, v in px.items() if not k.startswith("_")), key=lambda kv: -kv[1].get("GRBM_GUI_ACTIVE", 0)):
        if "GRBM_GUI_ACTIVE" in d and (d.get("SQ_INSTS_MFMA", 0) > 0 or "prep_i8" in name):
            lines.append(row(name + " (dist_only run, %s)" % what, d))
open(os.path.join(out, tag + "_derived.md"), "w").write("\n".join(lines) + "\n")

# the headline kernel: the packed-input instantiation when the bench ran it (its ASCII-resident twin runs in the same
# command for the `ascii_resident` object and gets its own file)
kms = sorted((k for k in pmc if k.startswith("kmer_sample_")), key=lambda k: -pmc[k].get("GRBM_GUI_ACTIVE", 0))
packed = [k for k in kms if k.startswith("kmer_sample_shared") and k.rstrip(">").split(",")[-1].strip() == "true"]
km = packed[0] if packed else kms[0]
for other in kms:
    if other != km and other.startswith("kmer_sample_shared"):
        json.dump({"kernel": other, "_stamp": pmc["_stamp"], "hbm_bytes_per_launch": pmc[other]["hbm_bytes_per_launch"],
                   "source": "as " + tag + "_kmer_traffic.json, for the ASCII-resident form of the same step"},
                  open(os.path.join(out, tag + "_kmer_ascii_traffic.json"), "w"), indent=1)
        break
json.dump({"kernel": km, "_stamp": pmc["_stamp"], "hbm_bytes_per_launch": pmc[km]["hbm_bytes_per_launch"],
           "hbm_read_bytes_per_launch_corrected": pmc[km]["hbm_read_bytes_per_launch_corrected"],
           "hbm_write_bytes_per_launch": pmc[km]["hbm_write_bytes_per_launch"],
           "source": "rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE (separate passes) of `python bench.py --steps 3 --warmup 1`; "
                     "FETCH_SIZE (KiB) x1024 x2 (gfx950 wide-read correction) + WRITE_SIZE (KiB) x1024, averaged per launch"},
          open(os.path.join(out, tag + "_kmer_traffic.json"), "w"), indent=1)
dm = sorted([k for k in pmcd if k.startswith("dist_mfma")], key=lambda k: -pmcd[k].get("GRBM_GUI_ACTIVE", 0))
if dm:
    d = pmcd[dm[0]]
    json.dump({"kernel": dm[0], "_stamp": pmcd["_stamp"], "hbm_bytes_per_launch": d["hbm_bytes_per_launch"],
               "l2_hit_rate": d["TCC_HIT_sum"] / (d["TCC_HIT_sum"] + d["TCC_MISS_sum"]),
               "mfma_busy": d["SQ_VALU_MFMA_BUSY_CYCLES"] / (SIMDS * d["GRBM_GUI_ACTIVE"] / 8.0),
               "source": "rocprofv3 --pmc passes of `python tools/dist_only.py` (10 000 x 10 000, ani_th 85); the bytes are "
                         "what leaves the XCD L2s (FETCH_SIZE x2 + WRITE_SIZE) -- the 168 MB of operands fit the 256 MiB "
                         "Infinity Cache, so most of it never reaches HBM"},
              open(os.path.join(out, tag + "_dist_traffic.json"), "w"), indent=1)
print(open(os.path.join(out, tag + "_derived.md")).read())
