"""bench.py contract on a tiny workload: one JSON line with the required keys (GPU)."""
import json
import os
import subprocess
import sys

import pytest

from conftest import ROOT

pytestmark = pytest.mark.gpu


def test_bench_json_contract():
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "1", "--steps", "2", "--warmup", "1",
                        "--genomes", "16", "--dist-n", "1024", "--hamming-refs", "4096", "--cpu-seconds", "0.5", "--small-genomes", "3000",
                        "--cli-sketch-files", "48"],
                       capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [l for l in r.stdout.splitlines() if l.strip()]
    assert len(lines) == 1
    j = json.loads(lines[0])
    for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling",
              "vs_baseline", "dtype", "data", "config", "roofline", "cpu_baseline"):
        assert k in j, k
    assert j["n_gpus"] == 1 and j["steps"] == 2 and j["warmup"] == 1 and j["value"] > 0
    assert j["vs_baseline"] is None and j["data"] == "synthetic" and "workload" in j["config"]
    rf = j["roofline"]
    assert rf["bound"] in ("hbm", "mfma") and abs(rf["frac"] - rf["achieved"] / rf["peak"]) < 1e-9 and rf["unit"] == "GB/s"
    cb = j["cpu_baseline"]
    assert cb["kind"] == "port" and cb["cores"] >= 1 and cb["value"] > 0 and "sample" in cb
    assert j["dist"]["roofline"]["bound"] == "mfma" and j["dist"]["value"] > 0
    assert j["hamming"]["value"] > 0
    # round 5: the same matrix with two distinct sets and with symmetric = 1; the inputs real users have; the CLI end to end
    d = j["dist"]
    assert d["two_sets"]["hits"] > 0 and d["two_sets"]["gemm_ms"] > 0 and d["symmetric"]["pairs"] == 1024 * 1023 // 2
    assert d["symmetric"]["hits"] * 2 + 1024 == d["config"]["hits_per_rank"]
    r = j["realistic"]
    assert r["draft_assemblies"]["value"] > 0 and r["draft_assemblies"]["ascii_resident"]["value"] > 0
    assert "CPU oracle" in r["draft_assemblies"]["parity"] and "CPU oracle" in r["many_small"]["parity"]
    assert r["many_small"]["value"] > 0 and 20 < r["many_small"]["nhash_mean"] < 50
    c = j["cli"]
    assert "error" not in c, c
    assert c["dist_symmetric"]["tsv_lines"] == d["symmetric"]["hits"] and c["dist_two_files"]["tsv_lines"] == d["two_sets"]["hits"]
    assert c["dist_two_files"]["wall_s"] > 0 and any(k.startswith("search") for k in c)
    assert c["sketch"]["files"] == 48 and c["sketch"]["files_per_s"] > 0 and c["sketch"]["sketch_file_mb"] > 0.1
    assert j["parity_gate"]["status"] == "passed" and j["parity_gate"]["ani_two_sets_hits_checked"] > 0


TWO_RANK_ARGS = ["--gpus", "2", "--steps", "2", "--warmup", "1", "--genomes", "24", "--genomes-10k", "50", "--dist-n", "2048",
                 "--hamming-refs", "6001", "--hamming-queries", "300", "--backend", "gloo", "--share-gpu"]


def _two_ranks(launcher):
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0")
    for k in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT"):
        env.pop(k, None)
    cmd = [sys.executable]
    if launcher:
        import socket
        s = socket.socket()
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
        s.close()
        cmd += ["-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
                "--master-port", str(port)]
    r = subprocess.run(cmd + [os.path.join(ROOT, "bench.py")] + TWO_RANK_ARGS, capture_output=True, text=True, timeout=900, env=env)
    assert r.returncode == 0, (r.stdout[-1500:], r.stderr[-3000:])
    lines = [l for l in r.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1  # rank 0 only
    return json.loads(lines[0])


def test_bench_two_ranks_share_one_gpu():
    """bench.py's N > 1 logic on a one-GPU box: two ranks (torch.distributed.run, gloo, both on device 0) run the real
    kernels through the sharded code paths -- weak-scaled sketch, the 10k-style strong-scaled leg, the all-gathered
    reference matrix of dist, the sharded Hamming search with ONE broadcast query set and merged hits (rank 0 verifies
    that every query finds its source row by global index).  RCCL itself is what the driver's 8-GPU run exercises."""
    j = _two_ranks(launcher=True)
    assert j["n_gpus"] == 2 and j["value"] > 0 and "cpu_baseline" not in j
    assert j["sketch_10k"]["config"]["genomes_per_gpu"] == 25 and j["sketch_10k"]["scaling"] == "strong"
    assert j["dist"]["value"] > 0 and j["dist"]["config"]["hits_per_rank"] > 0
    ex = j["dist"]["exchange"]  # prepared byte operands, chunked; checked against the i16 exchange inside the run
    assert ex["checked_against_i16_exchange"] and ex["fallbacks_to_i16"] == 0 and ex["bytes_per_rank_per_step"] < 0.6 * ex["i16_form_bytes"]
    assert j["hamming"]["config"]["hits_merged"] == 300 and j["hamming"]["config"]["refs_per_rank"] == 3001


def test_bench_bare_form_launches_its_own_ranks():
    """`python bench.py --gpus 2 ...` without torch.distributed.run in front (the form the driver uses at N = 1): the
    parent starts the two ranks as child processes and relays rank 0's line and the exit status."""
    j = _two_ranks(launcher=False)
    assert j["n_gpus"] == 2 and j["value"] > 0 and j["collectives"]["backend"] == "gloo" and j["collectives"]["world"] == 2
    assert j["dist"]["config"]["hits_per_rank"] > 0 and j["hamming"]["config"]["hits_merged"] == 300


def test_bench_rccl_one_rank():
    """RCCL's code path on the one-GPU box: `--backend nccl --collectives` initialises the process group through RCCL
    with ONE rank and runs the byte all-gather of the reference HV matrix, the query broadcast, the hit gather and the
    barrier / max-reduce brackets -- the same calls the 8-GPU run makes."""
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0")
    for k in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT"):
        env.pop(k, None)
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "1", "--steps", "2", "--warmup", "1",
                        "--genomes", "16", "--genomes-10k", "40", "--hostfed-genomes", "0", "--dist-n", "2048",
                        "--hamming-refs", "4096", "--hamming-queries", "300", "--backend", "nccl", "--collectives",
                        "--cpu-seconds", "0.5"], capture_output=True, text=True, timeout=900, env=env)
    assert r.returncode == 0, (r.stdout[-1500:], r.stderr[-3000:])
    j = json.loads([l for l in r.stdout.splitlines() if l.startswith("{")][0])
    assert j["collectives"] == dict(j["collectives"], backend="nccl", world=1)
    assert j["dist"]["config"]["hits_per_rank"] > 0 and j["hamming"]["config"]["hits_merged"] == 300
    assert j["parity_gate"]["status"] == "passed"  # the all-gathered matrix gave the same ANI block as the CPU
