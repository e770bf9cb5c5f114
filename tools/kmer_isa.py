#!/usr/bin/env python3
"""Static instruction budget of the canonical k = 21 sampling kernel (kmer_sample_shared<21, true, PACKED>) by class, per k-mer.

usage: tools/kmer_isa.py <tag> [packed]     writes profiles/<tag>_kmer_isa.txt and profiles/<tag>_kmer_isa.json
       (with `packed`: the hg_pack2-input instantiation, into profiles/<tag>_kmer_packed_isa.*)

The kernel is compiled to gfx950 assembly with the flags of hyper-gen_amd/csrc/Makefile and cut into basic blocks
(labels and branches).  Main path, per k-mer:
  * the k-mer body: one basic block per k-mer start j = 0..11 -- the strand compare of k-mer j + 1, the asm statement
    (word reads of k-mer j + 1, t1ha2 of k-mer j, threshold compare) and the scalar test of the hit mask; the twelve
    blocks of the clean-tile copy of the loop are averaged;
  * the tile level: staging (load, classify, the eight phase images, codes, validity word), the code window and the
    barriers, once per 12 k-mers and lane.
Not on the main path ("rare"): the u/U -> T rewrite (HG_NORM_U2T only), the per-base validity mask (tiles with a non-base
or a genome end), the second copy of the k-mer loop (same instructions, with the validity test in its hit path), hit
staging (1 k-mer in `scaled`), byte-wise tail loads, prologue / epilogue.
The per-k-mer budget is what bench.py's valu_issue object is priced with; the dynamic count (SQ_INSTS_VALU * 64 /
k-mers, profiles/<tag>_pmc.json) must agree with it up to the rare paths and the idle lanes (2 of 256 hash nothing).
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
PACKED = len(sys.argv) > 2 and sys.argv[2] == "packed"   # tools/kmer_isa.py <tag> packed: the hg_pack2-input instantiation
KERNEL = "kmer_sample_sharedILi21ELb1ELb%dEE" % (1 if PACKED else 0)
NAME = "kmer_sample_shared<21, true, %s>" % ("true" if PACKED else "false")
SUFFIX = "_kmer_packed_isa" if PACKED else "_kmer_isa"
M = 12            # k-mers per lane and tile

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
    tag = sys.argv[1] if len(sys.argv) > 1 else "r03"
    with tempfile.NamedTemporaryFile(suffix=".s") as f:
        subprocess.check_call(["hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-ffp-contract=off", "-mllvm",
                               "-amdgpu-atomic-optimizer-strategy=None", "-S", "--cuda-device-only", "-o", f.name, SRC],
                              stderr=subprocess.DEVNULL)
        lines = open(f.name).read().split("\n")
        if os.environ.get("HG_KEEP_ASM"):
            open(os.environ["HG_KEEP_ASM"], "w").write("\n".join(lines))
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
        cur.append((op, t))
        if op.startswith(("s_cbranch", "s_branch")):
            blocks.append((depth, cur))
            cur = []
    if cur:
        blocks.append((depth, cur))
    cnt = [collections.Counter(o for o, _ in b) for _, b in blocks]
    text = [" ".join(t for _, t in b) for _, b in blocks]
    hash_blocks = [i for i, c in enumerate(cnt) if c["v_mad_u64_u32"] >= 20]
    assert len(hash_blocks) == 2 * M, "expected two copies of the %d-k-mer loop, found %d hash blocks" % (M, len(hash_blocks))
    body = hash_blocks[:M]   # (the two copies' main-path blocks are the same instructions)
    md = blocks[hash_blocks[0]][0]  # loop depth of the tile loop (1 until round 5; 2 since the work items of a group are a loop around it)
    main_ops, rare_ops = collections.Counter(), collections.Counter()
    weights = collections.Counter()
    for i, ((d, b), c) in enumerate(zip(blocks, cnt)):
        in_loop_copy = min(hash_blocks) <= i <= max(hash_blocks)
        rare = (d != md or (in_loop_copy and i not in body)
                or any(c[o] for o in ("ds_add_rtn_u32", "global_atomic_add", "flat_store_dwordx2", "global_store_dwordx2",
                                      "global_load_ubyte", "s_endpgm"))
                or (c["v_mul_lo_u32"] >= 2 and c["v_mad_u64_u32"] == 0)                 # the per-base validity mask
                or ("0x55555555" in text[i] and c["v_perm_b32"] == 0 and i not in body))  # the u/U -> T rewrite
        if rare:
            rare_ops.update(c)
            continue
        per_kmer = 1.0 / M  # a body block is one of 12 k-mers; a tile-level block runs once per 12 k-mers
        weights["k-mer body" if i in body else "tile level"] += sum(n for o, n in c.items() if o.startswith("v_"))
        for o, n in c.items():
            main_ops[o] += n * per_kmer
    by_class = collections.Counter()
    for op, n in main_ops.items():
        by_class[cls(op)] += n
    valu = sum(n for op, n in main_ops.items() if op.startswith("v_"))
    slow = sum(n for k, n in by_class.items() if k.startswith(("slow", "multiply")))
    sys.path.insert(0, ROOT)
    import hypergen_amd as hg
    res = {"kernel": NAME, "source_sha": hg.source_stamp(), "kmers_per_lane_and_tile": M, "hash_blocks_found": len(hash_blocks),
           "static_valu_per_12_kmers": dict(weights),
           "per_kmer": {"valu": valu, "slow_class": slow, "plain": valu - slow,
                        "multiply": by_class["multiply (v_mad_u64_u32, v_mul_lo_u32)"],
                        "v_mov": by_class["plain: v_mov"],
                        "salu_and_other": sum(main_ops.values()) - valu},
           "rare_paths_static_valu": sum(n for op, n in rare_ops.items() if op.startswith("v_")),
           "by_class_per_kmer": {k: round(v, 3) for k, v in by_class.items()},
           "top_opcodes_per_kmer": {k: round(v, 3) for k, v in main_ops.most_common(40)},
           "method": "tools/kmer_isa.py: hipcc -S of hg_kmer_kernels.hip (Makefile flags); the twelve k-mer body blocks of one "
                     "copy of the loop + the tile-level blocks, per k-mer (see the tool's header)"}
    out = os.path.join(ROOT, "profiles")
    json.dump(res, open(os.path.join(out, tag + SUFFIX + ".json"), "w"), indent=1, sort_keys=True)
    with open(os.path.join(out, tag + SUFFIX + ".txt"), "w") as fo:
        fo.write("%s: static instruction budget of the main path, per k-mer\n" % NAME)
        fo.write("(static VALU per 12 k-mers and lane: %s)\n" % ", ".join("%s %d" % kv for kv in sorted(weights.items())))
        fo.write("VALU per k-mer %.1f = slow class %.1f (of which multiplies %.1f) + plain %.1f (of which v_mov %.1f)\n\n" % (
            valu, slow, res["per_kmer"]["multiply"], valu - slow, res["per_kmer"]["v_mov"]))
        fo.write("%-62s %10s\n" % ("class", "per k-mer"))
        for k, n in sorted(by_class.items(), key=lambda kv: -kv[1]):
            fo.write("%-62s %10.2f\n" % (k, n))
        fo.write("\n%-28s %10s   class\n" % ("opcode", "per k-mer"))
        for op, n in main_ops.most_common(60):
            fo.write("%-28s %10.2f   %s\n" % (op, n, cls(op)))
        fo.write("\nrare paths (u/U -> T rewrite, validity mask, second loop copy, hit staging, tails, prologue/epilogue): %d static VALU instructions\n"
                 % res["rare_paths_static_valu"])
    print(open(os.path.join(out, tag + SUFFIX + ".txt")).read())


if __name__ == "__main__":
    main()
