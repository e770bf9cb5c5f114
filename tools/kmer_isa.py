#!/usr/bin/env python3
"""Static instruction budget of the canonical k = 21 sampling kernel by class, per k-mer.

usage: tools/kmer_isa.py <tag>      writes profiles/<tag>_kmer_isa.txt and profiles/<tag>_kmer_isa.json
       HG_ISA_KERNEL=fast tools/kmer_isa.py <tag>_fast     the same for kmer_sample_fast<21, true, 28> (the A/B partner)

The kernel is compiled to gfx950 assembly with the flags of hyper-gen_amd/csrc/Makefile and cut into basic blocks
(labels and branches).  A block's weight is how often a lane runs it per k-mer on the main path:
  kmer_sample_grouped<21>: blocks of the tile loop (compiler's loop depth 1) run once per 36 k-mers, blocks of the group
      loop (depth 2: slice set-up, the 12 hash blocks, the slide to the next slice) once per 12;
  kmer_sample_fast<21, true, 28>: everything in the tile loop once per 12 k-mers.
Not on the main path ("rare"): the u/U -> T rewrite (HG_NORM_U2T only), the second copy of the 12 hash blocks (with the per-k-mer validity test: only waves
that see a non-base or a genome end), the validity-mask block, the byte-wise tail loads, hit staging (1 k-mer in
`scaled`), prologue / epilogue.
The per-k-mer budget is what bench.py's valu_issue object is priced with; the dynamic count (SQ_INSTS_VALU * 64 /
k-mers, profiles/<tag>_pmc.json) must agree with it up to the rare paths.
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
GROUPED = os.environ.get("HG_ISA_KERNEL", "grouped") == "grouped"
KERNEL = "kmer_sample_groupedILi21EE" if GROUPED else "kmer_sample_fastILi21ELb1ELi28EE"
NAME = "kmer_sample_grouped<21>" if GROUPED else "kmer_sample_fast<21, true, 28>"
M = 12            # k-mers per slice (hash blocks per copy of the k-mer loop)
GROUPS = 3 if GROUPED else 1

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
                               "-amdgpu-atomic-optimizer-strategy=None", "-S", "--cuda-device-only", "-o", f.name, SRC],
                              stderr=subprocess.DEVNULL)
        lines = open(f.name).read().split("\n")
    start = [i for i, l in enumerate(lines) if l.startswith("_Z") and ":" in l and KERNEL in l.split(":")[0]][0]
    end = start
    while "s_endpgm" not in lines[end]:
        end += 1
    # basic blocks: cut at labels and after branches; the loop depth comes from the compiler's label comments
    blocks, cur, depth = [], [], 0
    for l in lines[start + 1:end + 1]:
        t = l.strip()
        if not t:
            continue
        if t.startswith(".LBB"):
            if cur:
                blocks.append((depth, cur))
            cur = []
            m = re.search(r"Depth=(\d+)", t)
            depth = int(m.group(1)) if m else 0
            continue
        if t.startswith(";"):
            m = re.search(r"Depth=(\d+)", t)
            if m and not cur:
                depth = int(m.group(1))
            continue
        if t.startswith(".") or t.endswith(":"):
            continue
        op = t.split()[0]
        if not re.match(r"^[a-z_0-9]+$", op):
            continue
        cur.append(op)
        if op.startswith(("s_cbranch", "s_branch")):
            blocks.append((depth, cur))
            cur = []
    if cur:
        blocks.append((depth, cur))
    cnt = [collections.Counter(b) for _, b in blocks]
    hash_blocks = [i for i, c in enumerate(cnt) if c["v_mad_u64_u32"] >= 8]
    assert len(hash_blocks) == 2 * M, "expected two copies of the %d-k-mer loop, found %d hash blocks" % (M, len(hash_blocks))
    first, second = hash_blocks[:M], hash_blocks[M:]
    clean = first if sum(len(blocks[i][1]) for i in first) < sum(len(blocks[i][1]) for i in second) else second
    dirty = second if clean is first else first
    main_ops, rare_ops = collections.Counter(), collections.Counter()
    weights = collections.Counter()
    # the u/U -> T rewrite of the window (HG_NORM_U2T only, one uniform branch per window): the block(s) between the
    # window load and the classification block
    i_load = max(i for i, c in enumerate(cnt) if c["global_load_dwordx4"] and i < min(hash_blocks))
    i_cls = min(i for i, c in enumerate(cnt) if c["v_dot4_u32_u8"] >= 4)
    nv = [sum(n for o, n in c.items() if o.startswith("v_")) for c in cnt]
    first_big = min(i for i in range(i_load + 1, i_cls + 1) if nv[i] >= 24)
    u2t_blocks = set(range(i_load + 1, first_big + 1)) if first_big < i_cls else set()
    for i, ((d, b), c) in enumerate(zip(blocks, cnt)):
        rare = (i in dirty or i in u2t_blocks or d == 0 or any(c[o] for o in ("ds_add_rtn_u32", "global_atomic_add", "flat_store_dwordx2",
                                                          "global_store_dwordx2", "global_load_ubyte", "s_endpgm"))
                or (c["v_mul_lo_u32"] >= 6 and c["v_mad_u64_u32"] == 0)     # the per-base validity mask
                or (min(dirty) < i < max(dirty))                             # (its hit staging)
                or (min(clean) < i < max(clean) and i not in clean)          # hit staging between the hash blocks
                or (i > max(clean) and i > max(dirty) and c["v_alignbit_b32"] < 4)   # ... and after the last one
                or (max(clean) < i < min(dirty)))
        if GROUPED and not rare and d >= 3:
            rare = True                                                      # inner byte-wise loops
        if rare:
            rare_ops.update(c)
            continue
        per_kmer = 1.0 / M if (not GROUPED or d == 2) else 1.0 / (M * GROUPS)
        weights[round(1 / per_kmer)] += sum(n for o, n in c.items() if o.startswith("v_"))
        for o, n in c.items():
            main_ops[o] += n * per_kmer
    by_class = collections.Counter()
    for op, n in main_ops.items():
        by_class[cls(op)] += n
    valu = sum(n for op, n in main_ops.items() if op.startswith("v_"))
    slow = sum(n for k, n in by_class.items() if k.startswith(("slow", "multiply")))
    sys.path.insert(0, ROOT)
    import hypergen_amd as hg
    res = {"kernel": NAME, "source_sha": hg.source_stamp(), "kmers_per_slice": M, "slices_per_window": GROUPS, "hash_blocks_found": len(clean),
           "static_valu_by_period_in_kmers": {str(k): v for k, v in sorted(weights.items())},
           "per_kmer": {"valu": valu, "slow_class": slow, "plain": valu - slow,
                        "multiply": by_class["multiply (v_mad_u64_u32, v_mul_lo_u32)"],
                        "v_mov": by_class["plain: v_mov"],
                        "salu_and_other": sum(main_ops.values()) - valu},
           "rare_paths_static_valu": sum(n for op, n in rare_ops.items() if op.startswith("v_")),
           "by_class_per_kmer": {k: round(v, 3) for k, v in by_class.items()},
           "top_opcodes_per_kmer": {k: round(v, 3) for k, v in main_ops.most_common(40)},
           "method": "tools/kmer_isa.py: hipcc -S of hg_kmer_kernels.hip (Makefile flags); basic blocks weighted by how "
                     "often a lane runs them per k-mer on the main path (see the tool's header)"}
    out = os.path.join(ROOT, "profiles")
    json.dump(res, open(os.path.join(out, tag + "_kmer_isa.json"), "w"), indent=1, sort_keys=True)
    with open(os.path.join(out, tag + "_kmer_isa.txt"), "w") as fo:
        fo.write("%s: static instruction budget of the main path, per k-mer\n" % NAME)
        fo.write("(static VALU by how often a block runs: %s)\n" % ", ".join(
            "%d once per %s k-mers" % (v, k) for k, v in sorted(weights.items())))
        fo.write("VALU per k-mer %.1f = slow class %.1f (of which multiplies %.1f) + plain %.1f (of which v_mov %.1f)\n\n" % (
            valu, slow, res["per_kmer"]["multiply"], valu - slow, res["per_kmer"]["v_mov"]))
        fo.write("%-62s %10s\n" % ("class", "per k-mer"))
        for k, n in sorted(by_class.items(), key=lambda kv: -kv[1]):
            fo.write("%-62s %10.2f\n" % (k, n))
        fo.write("\n%-28s %10s   class\n" % ("opcode", "per k-mer"))
        for op, n in main_ops.most_common(60):
            fo.write("%-28s %10.2f   %s\n" % (op, n, cls(op)))
        fo.write("\nrare paths (validity mask, second loop copy, hit staging, tails, prologue/epilogue): %d static VALU instructions\n"
                 % res["rare_paths_static_valu"])
    print(open(os.path.join(out, tag + "_kmer_isa.txt")).read())


if __name__ == "__main__":
    main()
