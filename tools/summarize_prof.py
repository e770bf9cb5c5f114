#!/usr/bin/env python3
"""Condenses a tools/profile_gpu.sh output directory into small, committable summaries:
   <out>/<tag>_kernel_stats.csv   per-kernel calls / total / average ns (rocprofv3 --stats)
   <out>/<tag>_pmc.json           per-kernel counter sums per launch (+ corrected HBM traffic)
"""
import csv
import glob
import json
import os
import sys
from collections import defaultdict

OURS = ("prep_cen_kernel", "decide_cen_kernel", "kmer_sample", "sort_unique", "encode_wave_kernel", "encode_finalize_kernel", "encode_kernel", "bucket_count_kernel",
        "bucket_scan_kernel", "bucket_scatter_kernel", "bucket_sort_kernel", "bucket_copy_kernel", "dist_mfma", "dist_int",
        "prep_fast_kernel", "prep_i8_kernel", "i8_entries_kernel", "prep_kernel", "decide_kernel", "synth_kernel",
        "hamming_kernel", "binarize_kernel", "gather_keys_kernel", "permute_hits_kernel", "topk_kernel", "prep_cen_kernel",
        "decide_cen_kernel", "unpack_meta_kernel", "pack2_kernel", "expand_runs_kernel", "unpack2_kernel", "expand_bits_fp4_kernel",
        "expand_bits_kernel", "radix_hist_kernel", "radix_scan_kernel", "radix_scatter_kernel", "hg_hv_unpack_kernel")


def short(name):
    for k in OURS:
        if k in name:
            if "kmer_sample_fast" in name:
                return "kmer_sample_fast" + name.split("kmer_sample_fast")[1].split("(")[0]
            if "kmer_sample_grouped" in name:
                return "kmer_sample_grouped" + name.split("kmer_sample_grouped")[1].split("(")[0]
            if "kmer_sample_shared" in name:
                return "kmer_sample_shared" + name.split("kmer_sample_shared")[1].split("(")[0]
            if "kmer_sample_long" in name:
                return "kmer_sample_long" + name.split("kmer_sample_long")[1].split("(")[0]
            if "dist_mfma" in name:
                return "dist_mfma_kernel" + name.split("dist_mfma_kernel")[1].split("(")[0]
            return k
    return None


def main():
    out, tag = sys.argv[1], sys.argv[2]
    # ---- kernel stats
    rows = []
    for f in glob.glob(os.path.join(out, "stats", "**", "*kernel_stats.csv"), recursive=True):
        for r in csv.DictReader(open(f)):
            s = short(r["Name"])
            if s:
                rows.append((s, int(r["Calls"]), int(r["TotalDurationNs"]), float(r["AverageNs"]),
                             int(r["MinNs"]), int(r["MaxNs"])))
    rows.sort(key=lambda r: -r[2])
    with open(os.path.join(out, tag + "_kernel_stats.csv"), "w") as fo:
        fo.write("kernel,calls,total_ns,average_ns,min_ns,max_ns\n")
        for r in rows:
            fo.write('"%s",%d,%d,%.1f,%d,%d\n' % r)  # (template arguments carry commas)
            print("%-34s calls %3d  avg %10.1f us" % (r[0], r[1], r[3] / 1e3))
    # ---- counters: average per launch per kernel
    pmc = defaultdict(lambda: defaultdict(list))
    for f in glob.glob(os.path.join(out, "pmc_*", "**", "*counter_collection.csv"), recursive=True):
        for r in csv.DictReader(open(f)):
            s = short(r.get("Kernel_Name", ""))
            if s:
                pmc[s][r["Counter_Name"]].append(float(r["Counter_Value"]))
    summ = {}
    for k, d in pmc.items():
        summ[k] = {c: sum(v) / len(v) for c, v in d.items()}
        summ[k]["launches_seen"] = max(len(v) for v in d.values())
        if "FETCH_SIZE" in summ[k] or "WRITE_SIZE" in summ[k]:
            # FETCH_SIZE / WRITE_SIZE are in KiB; gfx950 counts wide coalesced reads at half their bytes
            # (MI355X_MICROARCH.md, HBM section) -> x2 on the read side
            fetch = summ[k].get("FETCH_SIZE", 0.0) * 1024.0 * 2.0
            write = summ[k].get("WRITE_SIZE", 0.0) * 1024.0
            summ[k]["hbm_read_bytes_per_launch_corrected"] = fetch
            summ[k]["hbm_write_bytes_per_launch"] = write
            summ[k]["hbm_bytes_per_launch"] = fetch + write
    # which tree these counters belong to (bench.py quotes them only for the same sources; tools/derive_prof.py adds the
    # commit when it copies the summary into profiles/)
    sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
    import hypergen_amd as hg
    summ["_stamp"] = {"source_sha": hg.source_stamp(), "kernels": sorted(k for k in summ if not k.startswith("_")),
                      "command": os.environ.get("HG_PROFILE_COMMAND", "")}
    json.dump(summ, open(os.path.join(out, tag + "_pmc.json"), "w"), indent=1, sort_keys=True)
    for k, d in summ.items():
        if not k.startswith("_"):
            print(k, json.dumps({c: round(v, 1) for c, v in d.items()}))


if __name__ == "__main__":
    main()
