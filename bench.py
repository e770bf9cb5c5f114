#!/usr/bin/env python3
"""bench.py -- HyperGen sketch + ANI hot path on MI355X.

    python bench.py --gpus N --steps K --warmup W
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...

One "step" = one pass of the sketch hot path (k-mer hash + FracMinHash sample -> set ->
HV encode -> norm; what src/sketch.rs:35-48 does per file) over one batch of synthetic genomes
that are already resident in HBM.  Workload at every N: BASELINE.json configs[1] per GPU
(1 000 synthetic 5 Mbp genomes, k=21 scaled=1500 D=4096) -- genomes shard embarrassingly, no
data-path collective, weak scaling.  The secondary metric (M ANI-pairs/s, configs[3]:
10 000 x 10 000 HVs, ani_th=85) is reported in the "dist" object of the same line; its only
exchange step is the RCCL all-gather of the reference HV matrix.

Prints ONE JSON line on rank 0.
"""
import argparse
import json
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)
import hypergen_amd as hg  # noqa: E402

L_GENOME = 5_000_000
HV_D = 4096
KSIZE = 21
SCALED = 1500
HBM_PEAK_GBS = 8000.0        # MI355X_MICROARCH.md: HBM3E 8 TB/s spec
MFMA_F16_PEAK_TFLOPS = 2500.0  # dense f16/bf16 MFMA peak


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=5)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--genomes", type=int, default=1000, help="genomes per GPU and step")
    ap.add_argument("--dist-n", type=int, default=10000, help="R = Q of the ANI matrix (0 = skip)")
    ap.add_argument("--hamming-refs", type=int, default=50000, help="refs of the bit-packed D=16384 search (0 = skip)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-seconds", type=float, default=12.0)
    return ap.parse_args()


def barrier_sync(world):
    torch.cuda.synchronize()
    if world > 1:
        torch.distributed.barrier()
    torch.cuda.synchronize()


def max_over_ranks(x, world, dev):
    if world == 1:
        return x
    t = torch.tensor([x], dtype=torch.float64, device=dev)
    torch.distributed.all_reduce(t, op=torch.distributed.ReduceOp.MAX)
    return float(t.item())


def clustered_hvs(rows, first_row, dev, n=3333, shared_frac=0.5, cluster=100):
    """Synthetic i16 HVs with the statistics of real sketches: hv = 2*count - n where count is
    Binomial(n, 1/2); members of a cluster share the counts of shared_frac*n hashes, so
    within-cluster ANI is ~96-97 % and cross-cluster ANI ~0 (about 1 % of pairs pass ani_th=85)."""
    ns = int(n * shared_frac)
    ids = torch.arange(first_row, first_row + rows, device=dev)
    cl = ids // cluster
    out = torch.empty((rows, HV_D), dtype=torch.int16, device=dev)
    half = torch.tensor(0.5, device=dev)
    for c in torch.unique(cl).tolist():
        g = torch.Generator(device=dev)
        g.manual_seed(0x48470000 + int(c))
        # normal approximation of Binomial(ns, 1/2), rounded: cheap, deterministic per cluster
        shared = torch.round(ns / 2 + (ns ** 0.5) / 2 * torch.randn(HV_D, generator=g, device=dev))
        m = cl == c
        k = int(m.sum())
        g2 = torch.Generator(device=dev)
        g2.manual_seed(0x48480000 + int(c) * 7919 + first_row)
        fresh = torch.round((n - ns) / 2 + ((n - ns) ** 0.5) / 2 * torch.randn((k, HV_D), generator=g2, device=dev))
        out[m] = (2 * (shared[None, :] + fresh) - n).to(torch.int16)
    del half
    return out


def valu_issue(kmer_ms, n_genomes):
    """Secondary, informative roofline of the k-mer kernel: its VALU instruction rate against the issue rate
    the same instruction mix reaches in tools/gpu_microbench.hip.  Instruction count and kernel cycles come
    from the committed PMC passes (profiles/r01_pmc.json: SQ_INSTS_VALU, GRBM_GUI_ACTIVE of this command)."""
    try:
        d = json.load(open(os.path.join(ROOT, "profiles", "r01_pmc.json")))
        k = [v for name, v in d.items() if name.startswith("kmer_sample_fast")][0]
        insts, cycles = k["SQ_INSTS_VALU"], k["GRBM_GUI_ACTIVE"] / 8.0
    except Exception:
        return None
    rate = insts / (1024 * cycles)  # wave-instructions per SIMD per cycle, as profiled
    # mix of the kernel: ~42 slow-class (3.9 cycles) + ~55 plain (2.3 cycles) instructions per k-mer
    # (profiles/r01_instruction_rates.txt) -> 97 instructions in ~290 cycles
    bound = 97.0 / (42 * 3.9 + 55 * 2.3)
    return {"valu_instr_per_simd_cycle": rate, "mix_issue_bound": bound, "frac": rate / bound,
            "valu_instr_per_kmer": insts * 64.0 / (n_genomes * (L_GENOME + 1 - KSIZE + 1)) if n_genomes == 1000 else None,
            "source": "profiles/r01_pmc.json (rocprofv3 --pmc of this command), profiles/r01_instruction_rates.txt"}


def effective_cores():
    """Host cores this process may really use: min(cpu_count, affinity mask, cgroup CPU quota),
    capped at 255 like the reference's `-t` (u8, src/utils.rs:54-56)."""
    n = os.cpu_count() or 1
    try:
        n = min(n, len(os.sched_getaffinity(0)))
    except Exception:
        pass
    try:
        quota, period = open("/sys/fs/cgroup/cpu.max").read().split()[:2]  # cgroup v2
        if quota != "max":
            n = min(n, max(1, int(int(quota) / int(period))))
    except Exception:
        try:
            q = int(open("/sys/fs/cgroup/cpu/cpu.cfs_quota_us").read())
            per = int(open("/sys/fs/cgroup/cpu/cpu.cfs_period_us").read())
            if q > 0:
                n = min(n, max(1, q // per))
        except Exception:
            pass
    return max(1, min(n, 255))


def cpu_baseline_sketch(seconds, log):
    """Oracle (CPU port of src/sketch.rs:35-56: per-genome task parallelism, ASCII canonical k-mers,
    t1ha2, set, AVX2-layout encode, norm) on a bounded sample of the same workload, all host cores."""
    from oracle import oracle as orc
    orc.lib()
    cores = effective_cores()
    n = min(max(2 * cores, 16), 512)                # bounded host memory: n x 5 MB
    genomes = orc.synth_genomes_mt(0, n, L_GENOME, cores)
    t0 = time.perf_counter()
    orc.sketch_batch_mt(genomes[:cores], cores)  # calibration pass (also warms the threads)
    t1 = time.perf_counter() - t0
    reps = max(1, min(20, int(seconds / max(t1 * n / cores, 1e-3))))
    t0 = time.perf_counter()
    for _ in range(reps):
        orc.sketch_batch_mt(genomes, cores)
    dt = time.perf_counter() - t0
    log("cpu baseline: %d genomes x %d passes on %d threads in %.1f s" % (n, reps, cores, dt))
    return {"value": n * reps / dt, "unit": "genomes/sec", "cores": cores, "kind": "port",
            "sample": "%d of the step's synthetic 5 Mbp genomes x %d passes, oracle/libhg_oracle.so "
                      "orc_sketch_batch_mt (CPU restatement of src/sketch.rs:35-56, OpenMP over genomes), "
                      "%d threads, %.1f s" % (n, reps, cores, dt)}


def cpu_baseline_dist(hv, n2, seconds, log):
    from oracle import oracle as orc
    cores = effective_cores()
    orc.set_threads(cores)
    rows = min(hv.shape[0], 512)
    r, rn = hv[:rows].cpu().numpy(), n2[:rows].cpu().numpy()
    t0 = time.perf_counter()
    orc.ani_matrix(r, rn, r, rn, KSIZE)
    t1 = time.perf_counter() - t0
    # grow the sub-block until it takes a few seconds (pairs scale with R x Q)
    q_rows = int(min(hv.shape[0], max(rows, rows * seconds / max(t1, 1e-4))))
    rows = min(hv.shape[0], 2048)
    r, rn = hv[:rows].cpu().numpy(), n2[:rows].cpu().numpy()
    q, qn = hv[:q_rows].cpu().numpy(), n2[:q_rows].cpu().numpy()
    t0 = time.perf_counter()
    orc.ani_matrix(r, rn, q, qn, KSIZE)
    dt = time.perf_counter() - t0
    log("cpu dist baseline: %d x %d in %.2f s" % (rows, q_rows, dt))
    return {"value": rows * q_rows / dt / 1e6, "unit": "M ANI-pairs/sec", "cores": cores, "kind": "port",
            "sample": "%d x %d sub-block of the step's HVs, oracle orc_ani_matrix (src/dist.rs:139-161, scalar "
                      "i16 dot per pair), OpenMP over rows on %d threads, %.2f s" % (rows, q_rows, cores, dt)}


def main():
    a = parse()
    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if world != a.gpus:
        if world == 1 and a.gpus > 1:
            raise SystemExit("launch with torch.distributed.run --nproc-per-node %d" % a.gpus)
    torch.cuda.set_device(local)
    dev = torch.device("cuda", local)
    if world > 1:
        torch.distributed.init_process_group("nccl", device_id=dev)

    def log(msg):
        if rank == 0:
            print("[bench] " + msg, file=sys.stderr, flush=True)

    ctx = hg.Context(local)
    ctx.set_stream(torch.cuda.current_stream().cuda_stream)

    # ---------------- sketch: genomes resident in HBM -------------------------------------------
    N = a.genomes
    stride = (L_GENOME + 1 + 15) // 16 * 16
    seq = torch.empty(N * stride + 64, dtype=torch.uint8, device=dev)
    ctx.synth_genomes_dev(rank * N, N, L_GENOME, stride, seq.data_ptr())
    offs = np.arange(N, dtype=np.uint64) * stride
    lens = np.full(N, L_GENOME + 1, np.uint64)
    p = hg.default_params(ksize=KSIZE, scaled=SCALED, hv_d=HV_D)
    hv = torch.empty((N, HV_D), dtype=torch.int16, device=dev)
    n2 = torch.empty(N, dtype=torch.int32, device=dev)
    nh = torch.empty(N, dtype=torch.int32, device=dev)

    def step():
        ctx.sketch_batch_dev(seq.data_ptr(), offs, lens, p, hv.data_ptr(), n2.data_ptr(), nh.data_ptr())

    for _ in range(a.warmup):
        step()
    ctx.enable_timing(True)
    ctx.timings()
    barrier_sync(world)
    t0 = time.perf_counter()
    for _ in range(a.steps):
        step()
    barrier_sync(world)
    dt = max_over_ranks(time.perf_counter() - t0, world, dev)
    tm = ctx.timings()
    ctx.enable_timing(False)
    kmer_ms, kmer_launches = tm["kmer"]
    kmer_avg_ms = kmer_ms / max(kmer_launches, 1)
    bytes_per_launch = N * (L_GENOME + 2 * HV_D)  # L + 2*D algorithmic bytes per genome
    achieved = bytes_per_launch / (kmer_avg_ms * 1e-3) / 1e9
    traffic = None
    tpath = os.path.join(ROOT, "profiles", "r01_kmer_traffic.json")
    if os.path.exists(tpath):
        try:
            traffic = json.load(open(tpath)).get("hbm_bytes_per_launch")
        except Exception:
            traffic = None
    value = N * world * a.steps / dt
    log("sketch: %.1f genomes/s, kmer kernel %.3f ms/launch (%d launches), nhash mean %.1f" % (
        value, kmer_avg_ms, kmer_launches, nh.float().mean().item()))

    out = {
        "metric": "genomes/sec sketch (k=21,s=1500,D=4096)", "value": value, "unit": "genomes/sec",
        "n_gpus": world, "steps": a.steps, "warmup": a.warmup, "ms_per_step": dt / a.steps * 1e3,
        "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "u64",
        "data": "synthetic",
        "config": {"workload": "%d synthetic 5 Mbp genomes per GPU (BASELINE configs[1]), sketch k=21 "
                               "scaled=1500 seed=123 canonical D=4096 AVX2 layout, inputs resident in HBM" % N,
                   "genomes_per_gpu": N, "genome_bp": L_GENOME, "parallelism": "genome-sharded x%d, no collective" % world},
        "roofline": {"bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                     "frac": achieved / HBM_PEAK_GBS, "traffic": traffic,
                     "kernel": "kmer_sample_fast<21,true>", "launch_ms": kmer_avg_ms,
                     "algorithmic_bytes_per_launch": bytes_per_launch,
                     "note": "nominally a scan, so priced against HBM; the true binder is integer VALU issue "
                             "(~97 VALU instructions per k-mer, ~60 of them the t1ha2 hash; see valu_issue)",
                     "valu_issue": valu_issue(kmer_avg_ms, N),
                     "kmer_hashes_per_sec": N * (L_GENOME + 1 - KSIZE + 1) / (kmer_avg_ms * 1e-3)},
        "kernel_ms_per_step": {k: v[0] / max(a.steps, 1) for k, v in tm.items() if v[1]},
    }

    # ---------------- dist: R x Q ANI matrix, thresholded ------------------------------------------
    if a.dist_n:
        R = a.dist_n
        rows = R // world
        mine = clustered_hvs(rows, rank * rows, dev)
        mine_n2 = (mine.int() ** 2).sum(1).int()
        cap = max(1 << 20, rows * R // 20)
        hits = torch.empty(cap * 3, dtype=torch.int32, device=dev)  # hg_ani_hit = 12 bytes
        if world > 1:
            ref_all = torch.empty((R // world * world, HV_D), dtype=torch.int16, device=dev)
            ref_n2 = torch.empty(R // world * world, dtype=torch.int32, device=dev)
        found = 0

        def dstep():
            nonlocal found
            if world > 1:  # the path's one exchange step: all-gather the reference HV matrix (RCCL/xGMI)
                # as raw bytes: RCCL has no int16 datatype and the payload is opaque to the collective
                torch.distributed.all_gather_into_tensor(ref_all.view(torch.uint8).view(-1),
                                                         mine.view(torch.uint8).view(-1))
                torch.distributed.all_gather_into_tensor(ref_n2, mine_n2)
                r, rn, nr = ref_all, ref_n2, ref_all.shape[0]
            else:
                r, rn, nr = mine, mine_n2, rows
            found, _ = ctx.dist_dev(r.data_ptr(), rn.data_ptr(), nr, mine.data_ptr(), mine_n2.data_ptr(), rows,
                                    HV_D, KSIZE, False, 85.0, hits.data_ptr(), cap)

        for _ in range(max(a.warmup, 1)):
            dstep()
        ctx.enable_timing(True)
        ctx.timings()
        barrier_sync(world)
        t0 = time.perf_counter()
        for _ in range(a.steps):
            dstep()
        barrier_sync(world)
        ddt = max_over_ranks(time.perf_counter() - t0, world, dev)
        dtm = ctx.timings()
        ctx.enable_timing(False)
        pairs = (rows * world) * rows * world  # all ranks together cover R x Q
        gemm_ms = dtm["dist"][0] / max(dtm["dist"][1], 1)
        flops_per_launch = 2.0 * HV_D * (rows * world) * rows  # 2*D ops per pair
        ach = flops_per_launch / (gemm_ms * 1e-3) / 1e12
        dist_traffic = None  # bytes leaving the XCD L2s per launch (PMC, profiles/): 10 000 x 10 000 only
        dpath = os.path.join(ROOT, "profiles", "r01_dist_traffic.json")
        if os.path.exists(dpath) and rows * world == 10000 and world == 1:
            try:
                dist_traffic = json.load(open(dpath)).get("hbm_bytes_per_launch")
            except Exception:
                dist_traffic = None
        out["dist"] = {
            "metric": "M ANI-pairs/sec (D=4096, ani_th=85)", "value": pairs * a.steps / ddt / 1e6,
            "unit": "M ANI-pairs/sec", "ms_per_step": ddt / a.steps * 1e3, "scaling": "strong",
            "config": {"workload": "%d ref x %d query clustered synthetic HVs (BASELINE configs[3]), thresholded "
                                   "output" % (rows * world, rows * world), "hits_per_rank": int(found)},
            "roofline": {"bound": "mfma", "achieved": ach, "peak": MFMA_F16_PEAK_TFLOPS, "unit": "TFLOP/s",
                         "frac": ach / MFMA_F16_PEAK_TFLOPS, "traffic": dist_traffic, "kernel": "dist_mfma_kernel (f16)",
                         "launch_ms": gemm_ms, "algorithmic_flops_per_launch": flops_per_launch},
            "kernel_ms_per_step": {k: v[0] / max(a.steps, 1) for k, v in dtm.items() if v[1]},
        }
        log("dist: %.0f M pairs/s, gemm %.3f ms/launch = %.1f TFLOP/s, hits/rank %d" % (
            out["dist"]["value"], gemm_ms, ach, found))

    # ---------------- bit-packed D=16384 Hamming search (BASELINE configs[4], extension) -------------------
    if a.hamming_refs:
        HD, HQ = 16384, 1000
        refs = a.hamming_refs // world  # the reference database is sharded, the (small) query set replicated
        gen = torch.Generator(device=dev)
        gen.manual_seed(0x48480000 + rank)
        rb = torch.randint(-2**31, 2**31 - 1, (refs, HD // 32), dtype=torch.int32, device=dev, generator=gen)
        qb = rb[:HQ].clone()
        qb ^= (1 << torch.randint(0, 31, (HQ, HD // 32), device=dev, generator=gen)).int()  # 512 flipped bits
        hcap = 1 << 20
        hh = torch.empty(hcap * 3, dtype=torch.int32, device=dev)
        nfound = 0

        def hstep():
            nonlocal nfound
            nfound, _ = ctx.hamming_search_dev(rb.data_ptr(), refs, qb.data_ptr(), HQ, HD, 2000, hh.data_ptr(), hcap)

        hstep()
        ctx.enable_timing(True)
        ctx.timings()
        barrier_sync(world)
        t0 = time.perf_counter()
        for _ in range(a.steps):
            hstep()
        barrier_sync(world)
        hdt = max_over_ranks(time.perf_counter() - t0, world, dev)
        htm = ctx.timings()
        ctx.enable_timing(False)
        hms = htm["dist"][0] / max(htm["dist"][1], 1)
        out["hamming"] = {
            "metric": "M Hamming-pairs/sec (D=16384 bit-packed)", "value": refs * world * HQ * a.steps / hdt / 1e6,
            "unit": "M pairs/sec", "ms_per_step": hdt / a.steps * 1e3, "scaling": "strong",
            "config": {"workload": "%d ref x %d query sign-binarised D=16384 HVs, popcount(xor) <= 2000 (BASELINE "
                                   "configs[4]; extension, no reference counterpart)" % (refs * world, HQ),
                       "hits_per_rank": int(nfound)},
            "kernel_ms": hms, "word_ops_per_sec": refs * HQ * (HD / 32) / (hms * 1e-3),
        }
        log("hamming: %.0f M pairs/s, kernel %.3f ms, hits/rank %d" % (out["hamming"]["value"], hms, nfound))
        del rb, qb, hh

    # ---------------- CPU baseline (rank 0, single-GPU runs only) ----------------------------------
    if rank == 0 and world == 1 and not a.no_cpu_baseline:
        out["cpu_baseline"] = cpu_baseline_sketch(a.cpu_seconds, log)
        out["speedup_vs_cpu_baseline"] = value / out["cpu_baseline"]["value"]
        if a.dist_n:
            cb = cpu_baseline_dist(mine, mine_n2, min(a.cpu_seconds, 6.0), log)
            out["dist"]["cpu_baseline"] = cb
            out["dist"]["speedup_vs_cpu_baseline"] = out["dist"]["value"] / cb["value"]
    if world > 1:
        torch.distributed.barrier()
        torch.distributed.destroy_process_group()
    if rank == 0:
        print(json.dumps(out), flush=True)
    ctx.close()


if __name__ == "__main__":
    main()
