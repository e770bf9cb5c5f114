#!/usr/bin/env python3
"""profiles/README.md for one round, written from the committed summaries themselves (kernel averages from
<tag>_kernel_stats.csv / <tag>_dist_kernel_stats.csv, the bench line from <tag>_bench_line.json) -- no hand-typed numbers.
usage: tools/profiles_readme.py <tag>"""
import csv
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
P = os.path.join(ROOT, "profiles")
tag = sys.argv[1] if len(sys.argv) > 1 else "r04"


def table(name, top=14):
    path = os.path.join(P, name)
    if not os.path.exists(path):
        return ["(no `%s`)" % name]
    rows = list(csv.DictReader(open(path)))
    rows.sort(key=lambda r: -int(r["total_ns"]))
    out = ["| kernel | launches | average ms | fastest ms | slowest ms |", "|---|---|---|---|---|"]
    for r in rows[:top]:
        out.append("| `%s` | %s | %.3f | %.3f | %.3f |" % (r["kernel"], r["calls"], float(r["average_ns"]) / 1e6, int(r["min_ns"]) / 1e6,
                                                           int(r["max_ns"]) / 1e6))
    return out


L = ["# profiles/", "",
     "Round files are named `rNN_*`; earlier rounds' files stay for comparison.  Every `%s_*.json` carries a `_stamp`: sha256 over the" % tag,
     "library's sources (`hypergen_amd.source_stamp()`), the kernel names in the file, the profiled command and the commit the summary",
     "was copied in at; `bench.py` quotes a file's counters only for the same sources and the kernel the library reports it launched.",
     "**This file is generated** (`tools/profiles_readme.py %s`) from the summaries beside it." % tag, "",
     "## `%s_kernel_stats.csv` -- `rocprofv3 --kernel-trace --stats` of the bench command" % tag, "",
     "`rocprofv3 --kernel-trace --stats --output-format csv -- python3 bench.py --steps 3 --warmup 1 --no-cpu-baseline --genomes-10k 0",
     "--hostfed-genomes 0` (`tools/profile_gpu.sh %s`; only the headline shapes are launched, so that per-kernel averages are per shape;" % tag,
     "profiled passes run at a lower clock than un-profiled ones).", ""]
L += table(tag + "_kernel_stats.csv")
L += ["", "## `%s_dist_kernel_stats.csv` -- the dist kernels alone (`tools/profile_dist.sh %sdist`: `python3 tools/dist_only.py --reps 4`, 10 000 x 10 000)" % (tag, tag), ""]
L += table(tag + "_dist_kernel_stats.csv", 8)
if os.path.exists(os.path.join(P, tag + "_distf16_kernel_stats.csv")):
    L += ["", "## `%s_distf16_kernel_stats.csv` -- the same command on large sketches (`--nhash 6666` from round 5 on, `--nhash 10000` before: centred f16 operands)" % tag, ""]
    L += table(tag + "_distf16_kernel_stats.csv", 4)
for suffix, what in (("dist2", "R and Q in DIFFERENT buffers (`--sets two`: other members of the same clusters, both prepasses)"),
                     ("distsym", "one set with `symmetric = 1` (`--sets sym`: R (R - 1) / 2 pairs)")):
    if os.path.exists(os.path.join(P, tag + "_" + suffix + "_kernel_stats.csv")):
        L += ["", "## `%s_%s_kernel_stats.csv` -- the same command, %s" % (tag, suffix, what), ""]
        L += table(tag + "_" + suffix + "_kernel_stats.csv", 2)
bl = os.path.join(P, tag + "_bench_line.json")
if os.path.exists(bl):
    j = json.load(open(bl))
    rf = j["roofline"]
    L += ["", "## `%s_bench_line.json` -- `python bench.py --steps 20 --warmup 3` on the committed sources" % tag, "",
          "| | |", "|---|---|",
          "| sketch, packed bases resident | %.0f genomes/s, kernel `%s` %.3f ms / launch = %.4f of the HBM peak (algorithmic L + 2 D bytes) |" % (
              j["value"], rf["kernel"], rf["launch_ms"], rf["frac"]),
          "| sketch, ASCII resident | %.0f genomes/s, `%s` %.3f ms |" % (j["ascii_resident"]["value"], j["ascii_resident"]["kernel"], j["ascii_resident"]["launch_ms"])]
    if "sketch_10k" in j:
        L.append("| 10 000 genomes on one GPU | %.0f genomes/s |" % j["sketch_10k"]["value"])
    if "host_fed" in j:
        hf = j["host_fed"]
        L.append("| host-fed (PCIe) | hg_sketch_batch %.0f genomes/s (%s; %.0f as ASCII); 2-bit packed stream %.0f (sparse form, %.3f B/base) / %.0f (bitmap form) |" % (
            hf["value"], hf.get("link_form", "ASCII"), hf.get("ascii_link", hf)["value"], hf["packed_stream"]["value"],
            hf["packed_stream"]["bytes_per_base"], hf["packed_stream"]["bitmap_form"]["value"]))
    if "per_call" in j:
        pc = j["per_call"]
        L.append("| one call per genome, %d host threads | hg_kmer_hash_sample %.0f genomes/s (%s; %.0f as ASCII), hg_sketch_batch(n=1) %.0f |" % (
            pc["threads"], pc["value"], pc.get("link_form", "ASCII"), pc.get("ascii_link", pc)["value"], pc["sketch_one"]["value"]))
    if "dist" in j:
        d = j["dist"]
        L.append("| dist 10 000 x 10 000 | %.0f M pairs/s, GEMM %.3f ms = %.3f of the %s peak |" % (
            d["value"], d["roofline"]["launch_ms"], d["roofline"]["frac"], d["roofline"]["peak_dtype"]))
        for key, name in (("two_sets", "two distinct sets"), ("symmetric", "symmetric = 1")):
            if key in d:
                L.append("| dist, %s | %.0f M pairs/s, GEMM %.3f ms = %.3f of the peak, prepass %.3f ms, %d hits |" % (
                    name, d[key]["value"], d[key]["gemm_ms"], d[key]["frac_of_peak"], d[key]["prep_ms"], d[key]["hits"]))
    if "hamming" in j:
        h = j["hamming"]
        L.append("| Hamming search 50 000 x 10 000 x 16384 | %.0f M pairs/s, kernel %.3f ms = %.3f of the %s peak |" % (
            h["value"], h["kernel_ms"], h["roofline"]["frac"], h["roofline"].get("peak_dtype", "")))
    if "realistic" in j:
        r = j["realistic"]
        da = r["draft_assemblies"]
        L.append("| draft assemblies (200 contigs, soft-masked, IUPAC, tandem repeat) | %.0f genomes/s packed = %.3f of clean, %.0f ASCII |" % (
            da["value"], da["vs_clean"], da["ascii_resident"]["value"]))
        if "many_small" in r:
            ms = r["many_small"]
            L.append("| 100 000 genomes of 50 kbp | %.0f genomes/s = %.0f Mbase/s (%.2f of the clean per-base rate); kernels %s |" % (
                ms["value"], ms["mbases_per_sec"], ms["vs_clean_per_base"], {k: round(v, 2) for k, v in ms["kernel_ms_per_step"].items()}))
    if "cli" in j:
        c = j["cli"]
        L.append("| `hyper-gen dist -r A -q A` / `-r A -q B` / `search`, 10 000 sketches, end to end | %.2f / %.2f / %.2f s wall (%.3f / %.3f / %.3f s as the tool reports) |" % (
            c["dist_symmetric"]["wall_s"], c["dist_two_files"]["wall_s"], [v for k, v in c.items() if k.startswith("search")][0]["wall_s"],
            c["dist_symmetric"].get("reported_s", 0), c["dist_two_files"].get("reported_s", 0), [v for k, v in c.items() if k.startswith("search")][0].get("reported_s", 0)))
L += ["", "## the other files", "",
      "| file | what | made by |", "|---|---|---|",
      "| `%s_rocprofv3_kernel_stats_full.csv` | the unfiltered `--stats` table of the bench command | `tools/profile_gpu.sh %s` |" % (tag, tag),
      "| `%s_pmc.json`, `%s_dist_pmc.json` | per-kernel counter averages per launch from the separate `--pmc` passes (FETCH_SIZE; WRITE_SIZE; SQ_*; GRBM / MFMA / LDS; TCC) + corrected HBM bytes | `tools/summarize_prof.py` |" % (tag, tag),
      "| `%s_derived.md` | per-kernel derived metrics (instr / cycle, MFMA busy, LDS busy, LDS conflicts, L2 hit, wait split, HBM bytes) | `tools/derive_prof.py %s` |" % (tag, tag),
      "| `%s_kmer_traffic.json`, `%s_kmer_ascii_traffic.json`, `%s_dist_traffic.json` | the `roofline.traffic` figures `bench.py` reports | idem |" % (tag, tag, tag),
      "| `%s_kmer_packed_isa.*`, `%s_kmer_isa.*` | static instruction budget of `kmer_sample_shared<21, true, PACKED>` by class, per k-mer | `tools/kmer_isa.py %s [packed]` |" % (tag, tag, tag),
      "| `r06_dist_epilogue.md` (+ `r06_dist_epilogue_raw.txt`) | where the dist kernel's time goes between \"no candidates\" and 1.29 M hits: every tile split into main loop / phase 0 / append / phase 2 / reservation / hit write from stamps of all 1 280 workgroups, A/B builds of the phase-2 arithmetic, the sixth round of 47 left-over tiles | `tools/dist_epilogue_split.py` on `-DHG_DIST_STAMPS` builds (`tools/build_variant.sh`) |",
      "| `r06_small_and_long_k.txt` | the small-genome ladder (400 000 x 2 kbp .. 1 000 x 5 Mbp: Mbase/s and kernel times per step) and the k-mer kernel's time at k = 21 .. 255, with what changed in round 6 | `tools/small_genome_sweep.sh`, `tools/kmer_time.py` |",
      "| `r06_realistic_kernel_stats.txt` | `rocprofv3 --kernel-trace --stats` of the clean / draft-assembly / many-small-genomes sketch step, both resident forms (round 6: the step's new kernels -- `sort_unique_wave_kernel`, `sort_unique_rest_kernel`, `sketch_finish_kernel`) | `tools/profile_realistic.sh r06` |",
      "| `design_r05_full.md` | DESIGN.md as it stood at the end of round 5 | -- |",
      "| `r05_gemm_bounds.md` (+ `r05_l2_bound.txt`, `r05_mfma_shape.txt`, `r05_mfma_loop.txt`, `r05_epilogue_bounds.txt`) | what bounds the three GEMM kernels: the all-L2-hits build, MFMA shapes and operand values, the loop's ingredients, the uneven loader split, the epilogue's removable work | `tools/dist_only.py` on `-DHG_DIST_EXPERIMENT` builds, `tools/mfma_shape_microbench.hip`, `tools/mfma_microbench.hip` |",
      "| `r05_realistic_kernel_stats.txt` | `rocprofv3 --kernel-trace --stats` of the clean / draft-assembly / many-small-genomes sketch step, both resident forms | `tools/profile_realistic.sh r05` |",
      "| `r05_cli_kernel_stats.txt` | `rocprofv3 --kernel-trace --stats` of the CLI binary itself (`hyper-gen dist` / `search` on two 10 000-sketch files): every kernel an end-to-end comparison launches | `tools/profile_cli.sh r05` |",
      "| `r05_cli_dist.txt` | `hyper-gen dist` / `search` end to end with the tool's own stage timings, at the start and at the end of round 5 | `tools/cli_dist_bench.py` |",
      "| `design_r04_full.md` | DESIGN.md as it stood at the end of round 4 (every round-1..4 narrative) | -- |",
      "| `r04_dist_tile_table.txt` | per-CU timelines of one dist launch (which CU ran which workgroups, from whole-tile stamps + `HW_ID`) with the blockIdx mapping and with the host-built slot -> tile table, and the timings of both | `tools/dist_cu_timeline.py` on a `-DHG_DIST_STAMPS` build, `tools/dist_only.py` |",
      "| `r04_hostfed_probe.txt` | the one-call-per-genome pattern by number of calling threads and link form (ASCII / packed / the library's choice), and its ingredients alone (host packing, resident calls, uploads) | `tools/percall_probe.py`, `tools/percall_ingredients.py` |",
      "| `r04_dist_defer_neutral.txt` | deferred candidate evaluation (tiles append candidates, a second kernel evaluates them): kernel times, per-CU spans and tile durations against in-tile evaluation | `tools/dist_only.py`, `tools/dist_cu_timeline.py` |",
      "| `r04_dist_half_slots_negative.txt` | half tiles at the end of the XCDs' queues: timings against whole tiles, per-CU spans | `tools/dist_only.py`, `tools/dist_cu_timeline.py` |",
      "| `r04_dist_persistent_negative.txt` | the persistent-workgroup GEMM variant of round 4: timings against one tile per workgroup and whole-tile `s_memtime` stamps of both | `tools/dist_only.py`, `tools/dist_tile_stamps.py` on `-DHG_DIST_STAMPS` builds |",
      "| `r03_mfma_ceiling.txt` | what the matrix pipe sustains in the GEMM's loop shape, ingredient by ingredient | `tools/mfma_microbench.hip` |",
      "| `r01_instruction_rates.txt` | measured issue cost of the integer instructions the k-mer kernel is made of | `tools/gpu_microbench.hip` |"]
open(os.path.join(P, "README.md"), "w").write("\n".join(L) + "\n")
print("\n".join(L))
