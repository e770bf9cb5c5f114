#!/usr/bin/env python3
"""Static instruction budget of kmer_sample_fast<21, true> by class, per k-mer.

usage: tools/kmer_isa.py <tag>      writes profiles/<tag>_kmer_isa.txt and profiles/<tag>_kmer_isa.json

The kernel is compiled to gfx950 assembly with the flags of hyper-gen_amd/csrc/Makefile.  Its loop body
processes one 32-base window per lane = M = 12 k-mer starts.  Basic blocks are classified as
  main : executed by every lane for every tile -- the window load + SWAR classification block (holds the
         v_dot4_u32_u8 instructions) and the 12 hash blocks (hold the v_mad_u64_u32 instructions);
  rare : everything else inside the kernel -- the per-base validity mask (only waves that see a non-base or a
         genome end), hit staging (1 k-mer in `scaled`), spill path, prologue / epilogue.
"main" / 12 is the per-k-mer budget bench.py's valu_issue object is priced with; the dynamic count
(SQ_INSTS_VALU * 64 / k-mers, profiles/<tag>_pmc.json) must agree with it up to the rare paths.
Instruction classes follow the measured issue costs (profiles/r01_instruction_rates.txt): "slow" = ~3.7-4.1
cycles per wave-instruction (multiplies, 64-bit shifts / adds / compares, carry ops, v_perm / v_alignbyte, the
VOP3 forms v_add3 / v_lshl_or / v_and_or / v_bfi / v_bfe / v_cndmask_e64), "plain" = ~2.2-2.4 cycles.
"""
import collections
import json
import os
import re
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SRC = os.path.join(ROOT, "hyper-gen_amd", "csrc", "hg_kmer_kernels.hip")
KERNEL = "kmer_sample_fastILi21ELb1ELi%sE" % os.environ.get("HG_ISA_VAR", "28")  # <K = 21, CANON = true, VAR = HG_KMER_DEFAULT_VAR>
M = 12

SLOW = ("v_mad_u64_u32", "v_mul_lo_u32", "v_mul_hi_u32", "v_mul_u32_u24", "v_perm_b32", "v_alignbyte_b32",
        "v_alignbit_b32", "v_lshl_add_u64", "v_lshrrev_b64", "v_lshlrev_b64", "v_ashrrev_i64", "v_add3_u32",
        "v_lshl_or_b32", "v_and_or_b32", "v_bfi_b32", "v_bfe_u32", "v_cndmask_b32_e64", "v_add_co_u32",
        "v_addc_co_u32", "v_sub_co_u32", "v_subb_co_u32", "v_dot4_u32_u8", "v_mad_u32_u24", "v_xad_u32",
        "v_add_lshl_u32", "v_lshl_add_u32", "v_or3_b32", "v_xor3_b32")
MUL = ("v_mad_u64_u32", "v_mul_lo_u32", "v_mul_hi_u32")


def cls(op):
    if op.startswith("ds_"):
        return "LDS (issues beside the VALU stream)"
    if not op.startswith("v_"):
        return "scalar / memory / other"
    if op.startswith(MUL):
        return "multiply (v_mad_u64_u32, v_mul_lo_u32)"
    if op.startswith("v_cmp") and ("_u64" in op or "_i64" in op):
        return "slow: 64-bit compare"
    if op.startswith(SLOW):
        return "slow: 64-bit shift/add, carry, perm/alignbyte, VOP3 3-input"
    if op.startswith("v_mov"):
        return "plain: v_mov"
    return "plain: logic / shift / add / bitop3 / compare"


def main():
    tag = sys.argv[1] if len(sys.argv) > 1 else "r02"
    with tempfile.NamedTemporaryFile(suffix=".s") as f:
        subprocess.check_call(["hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-ffp-contract=off", "-mllvm",
                               "-amdgpu-atomic-optimizer-strategy=None", "-DHG_KMER_EXPERIMENT", "-S", "--cuda-device-only", "-o", f.name, SRC],
                              stderr=subprocess.DEVNULL)
        lines = open(f.name).read().split("\n")
    start = [i for i, l in enumerate(lines) if l.startswith("_Z") and ":" in l and KERNEL in l.split(":")[0]][0]
    end = start
    while "s_endpgm" not in lines[end]:
        end += 1
    # split into basic blocks at labels and after branches
    blocks, cur = [], []
    for l in lines[start + 1:end + 1]:
        t = l.strip()
        if not t or t[0] in ";" or t.startswith((".", ";;")) and not t.startswith(".LBB"):
            continue
        if t.startswith(".LBB") or t.endswith(":"):
            if cur:
                blocks.append(cur)
            cur = []
            continue
        op = t.split()[0]
        if not re.match(r"^[a-z_0-9]+$", op):
            continue
        cur.append(op)
        if op.startswith(("s_cbranch", "s_branch")):
            blocks.append(cur)
            cur = []
    if cur:
        blocks.append(cur)
    main_ops, rare_ops = collections.Counter(), collections.Counter()
    hash_blocks = [i for i, b in enumerate(blocks) if collections.Counter(b)["v_mad_u64_u32"] >= 8]
    # VAR & 16 compiles the k-mer loop twice (with and without the per-k-mer validity test): the main path is the
    # copy with fewer instructions, the other one belongs to the rare paths
    if len(hash_blocks) == 2 * M:
        first, second = hash_blocks[:M], hash_blocks[M:]
        hash_blocks = first if sum(len(blocks[i]) for i in first) < sum(len(blocks[i]) for i in second) else second
    n_hash_blocks = len(hash_blocks)
    for i, b in enumerate(blocks):
        c = collections.Counter(b)
        is_classify = c["v_dot4_u32_u8"] >= 4
        (main_ops if (i in hash_blocks or is_classify) else rare_ops).update(c)
    by_class = collections.Counter()
    for op, n in main_ops.items():
        by_class[cls(op)] += n
    valu = sum(n for op, n in main_ops.items() if op.startswith("v_"))
    slow = sum(n for k, n in by_class.items() if k.startswith(("slow", "multiply")))
    res = {"kernel": "kmer_sample_fast<21, true>", "kmers_per_lane_and_tile": M, "hash_blocks_found": n_hash_blocks,
           "main_path_static": {"valu": valu, "slow_class": slow, "plain": valu - slow,
                                "salu_and_other": sum(main_ops.values()) - valu},
           "per_kmer": {"valu": valu / M, "slow_class": slow / M, "plain": (valu - slow) / M,
                        "multiply": by_class["multiply (v_mad_u64_u32, v_mul_lo_u32)"] / M,
                        "v_mov": by_class["plain: v_mov"] / M},
           "rare_paths_static_valu": sum(n for op, n in rare_ops.items() if op.startswith("v_")),
           "by_class_main_path": dict(by_class),
           "top_opcodes_main_path": dict(main_ops.most_common(40)),
           "method": "tools/kmer_isa.py: hipcc -S of hg_kmer_kernels.hip (Makefile flags); basic blocks holding >= 8 "
                     "v_mad_u64_u32 (12 hash blocks) or >= 4 v_dot4_u32_u8 (classification) are the main path"}
    out = os.path.join(ROOT, "profiles")
    json.dump(res, open(os.path.join(out, tag + "_kmer_isa.json"), "w"), indent=1, sort_keys=True)
    with open(os.path.join(out, tag + "_kmer_isa.txt"), "w") as fo:
        fo.write("kmer_sample_fast<21, true>: static instruction budget of the main path (%d hash blocks + classification)\n" % n_hash_blocks)
        fo.write("VALU per k-mer %.1f = slow class %.1f (of which multiplies %.1f) + plain %.1f (of which v_mov %.1f)\n\n" % (
            valu / M, slow / M, res["per_kmer"]["multiply"], (valu - slow) / M, res["per_kmer"]["v_mov"]))
        fo.write("%-62s %8s %10s\n" % ("class", "static", "per k-mer"))
        for k, n in sorted(by_class.items(), key=lambda kv: -kv[1]):
            fo.write("%-62s %8d %10.2f\n" % (k, n, n / M))
        fo.write("\n%-28s %8s %10s   class\n" % ("opcode", "static", "per k-mer"))
        for op, n in main_ops.most_common(60):
            fo.write("%-28s %8d %10.2f   %s\n" % (op, n, n / M, cls(op)))
        fo.write("\nrare paths (validity mask, hit staging, spill, prologue/epilogue): %d static VALU instructions\n" % res["rare_paths_static_valu"])
    print(open(os.path.join(out, tag + "_kmer_isa.txt")).read())


if __name__ == "__main__":
    main()
