"""world_size-2 CPU (gloo) test of the multi-GPU decomposition used by bench.py: genome sharding
without a collective, one all-gather of the reference HVs for dist."""
import os
import socket
import sys

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from conftest import ROOT


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, n_genomes, out_dir):
    sys.path.insert(0, ROOT)
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    import hypergen_amd  # noqa: F401  (package import only; no device needed)
    from hypergen_amd import shard
    from oracle import oracle as orc
    lo, hi = shard.shard_range(n_genomes, rank, world)
    # "sketch" shard: every rank sketches only its genomes (CPU oracle stands in for the kernel here)
    hv = np.stack([orc.sketch_genome(orc.synth_genome(g, 30_000), scaled=20, hv_d=512)[0] for g in range(lo, hi)])
    n2 = np.array([orc.hv_norm2(r) for r in hv], np.int32)
    # dist: all-gather the reference HV matrix, compute this rank's R x Q_local block
    ref = shard.allgather_rows(torch.from_numpy(hv), world).numpy()
    ref_n2 = shard.allgather_rows(torch.from_numpy(n2), world).numpy()
    # the exact call pattern of bench.py's dist step for equal-sized shards: int16 rows gathered as raw
    # bytes into a preallocated tensor (RCCL, like gloo, has no int16 datatype)
    eq = torch.from_numpy(hv[:2].copy())
    ref_eq = torch.empty((2 * world, eq.shape[1]), dtype=torch.int16)
    dist.all_gather_into_tensor(ref_eq.view(torch.uint8).view(-1), eq.view(torch.uint8).view(-1))
    assert (ref_eq[2 * rank:2 * rank + 2] == eq).all()
    block = orc.ani_matrix(ref, ref_n2, hv, n2, 21)
    np.save(os.path.join(out_dir, "block%d.npy" % rank), block)
    np.save(os.path.join(out_dir, "hv%d.npy" % rank), hv)
    dist.barrier()
    dist.destroy_process_group()


def _search_worker(rank, world, port, n_refs, n_qry, words, max_dist, out_dir):
    """configs[4] decomposition: refs sharded by rows, ONE query set broadcast from rank 0, hits merged."""
    sys.path.insert(0, ROOT)
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    import hypergen_amd as hg
    from hypergen_amd import shard
    from oracle import oracle as orc
    rng = np.random.default_rng(1234)            # the database exists "on disk": every rank cuts its own rows
    refs = rng.integers(0, 2**32, (n_refs, words), dtype=np.uint64).astype(np.uint32)
    lo, hi = shard.shard_range(n_refs, rank, world)
    # only rank 0 knows the queries; the other ranks start from garbage that the broadcast must replace
    if rank == 0:
        q = refs[rng.integers(0, n_refs, n_qry)].copy()
        q ^= (np.uint32(1) << rng.integers(0, 32, q.shape).astype(np.uint32))
        np.save(os.path.join(out_dir, "queries.npy"), q)
    else:
        q = np.full((n_qry, words), 0xDEADBEEF, np.uint32)
    qt = torch.from_numpy(q)

    def search_block(ref_local, ref_lo, queries):  # the CPU oracle stands in for hg_hamming_search_block_dev
        d = orc.hamming_matrix(ref_local, queries.numpy())
        ri, qi = np.nonzero(d <= max_dist)
        h = np.zeros(ri.size, hg.HAM_HIT_DTYPE)
        h["ref_idx"], h["qry_idx"], h["dist"] = ri + ref_lo, qi, d[ri, qi]
        return h

    merged = shard.sharded_search(search_block, refs[lo:hi], lo, qt, world)
    np.save(os.path.join(out_dir, "merged%d.npy" % rank), merged)
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.timeout(300)
def test_two_rank_sharded_search_broadcasts_queries_and_merges_hits(tmp_path, orc):
    import hypergen_amd as hg
    n_refs, n_qry, words, max_dist, world = 301, 23, 16, 40, 2
    mp.spawn(_search_worker, args=(world, _free_port(), n_refs, n_qry, words, max_dist, str(tmp_path)), nprocs=world,
             join=True)
    rng = np.random.default_rng(1234)
    refs = rng.integers(0, 2**32, (n_refs, words), dtype=np.uint64).astype(np.uint32)
    q = np.load(tmp_path / "queries.npy")
    d = orc.hamming_matrix(refs, q)               # single-process oracle on the whole database
    ri, qi = np.nonzero(d <= max_dist)
    want = sorted(zip(ri.tolist(), qi.tolist(), d[ri, qi].tolist()))
    assert len(want) >= n_qry                     # every query finds at least its source row
    for r in range(world):                        # every rank holds the same merged list
        m = np.load(tmp_path / ("merged%d.npy" % r))
        assert m.dtype == hg.HAM_HIT_DTYPE
        assert sorted(zip(m["ref_idx"].tolist(), m["qry_idx"].tolist(), m["dist"].tolist())) == want


def test_hg_shard_range_matches_the_python_rule():
    sys.path.insert(0, ROOT)
    import hypergen_amd as hg
    from hypergen_amd import shard
    for n in (0, 1, 7, 8, 1000, 10001):
        for w in (1, 2, 3, 8):
            for k in range(w):
                assert hg.shard_range(n, k, w) == shard.shard_range(n, k, w)


def test_shard_range_covers_everything():
    sys.path.insert(0, ROOT)
    import hypergen_amd  # noqa: F401
    from hypergen_amd import shard
    for n in (0, 1, 7, 8, 1000, 10001):
        for w in (1, 2, 3, 8):
            r = [shard.shard_range(n, k, w) for k in range(w)]
            assert r[0][0] == 0 and r[-1][1] == n and all(a[1] == b[0] for a, b in zip(r, r[1:]))
            assert max(b - a for a, b in r) - min(b - a for a, b in r) <= 1


@pytest.mark.timeout(300)
def test_two_rank_sketch_and_dist_decomposition(tmp_path, orc):
    n, world = 5, 2  # odd on purpose: ranks own 3 and 2 genomes
    mp.spawn(_worker, args=(world, _free_port(), n, str(tmp_path)), nprocs=world, join=True)
    hv = np.concatenate([np.load(tmp_path / ("hv%d.npy" % r)) for r in range(world)])
    want_hv = np.stack([orc.sketch_genome(orc.synth_genome(g, 30_000), scaled=20, hv_d=512)[0] for g in range(n)])
    assert (hv == want_hv).all()  # sharded sketches == single-process sketches, in genome order
    n2 = np.array([orc.hv_norm2(r) for r in hv], np.int32)
    full = orc.ani_matrix(hv, n2, hv, n2, 21)
    got = np.concatenate([np.load(tmp_path / ("block%d.npy" % r)) for r in range(world)], axis=1)
    assert got.shape == full.shape and (got == full).all()


def test_bench_self_launch_parent_stays_off_the_gpu():
    """`python bench.py --gpus N` (N > 1, no torch.distributed environment) must start its ranks as CHILD processes
    before torch or the HIP library is loaded, pass its own arguments on and exit with the children's status."""
    code = r"""
import os, subprocess, sys
for k in ("RANK", "WORLD_SIZE", "LOCAL_RANK"):
    os.environ.pop(k, None)
sys.path.insert(0, %r)
import bench
calls = []
class R:
    returncode = 7
def fake_run(cmd, **kw):
    calls.append((cmd, kw))
    return R()
subprocess.run = fake_run
sys.argv = ["bench.py", "--gpus", "4", "--steps", "3", "--warmup", "2"]
try:
    bench.main()
    code = None
except SystemExit as e:
    code = e.code
assert code == 7, code
assert "torch" not in sys.modules and "hypergen_amd" not in sys.modules and "numpy" not in sys.modules
(cmd, kw), = calls
i = cmd.index("--nproc-per-node")
assert cmd[1:3] == ["-m", "torch.distributed.run"] and cmd[i + 1] == "4" and "127.0.0.1" in cmd
assert cmd[-6:] == ["--gpus", "4", "--steps", "3", "--warmup", "2"] and cmd[-7].endswith("bench.py")
assert kw["env"]["HSA_ENABLE_IPC_MODE_LEGACY"] == "0"
print("ok")
""" % ROOT
    import subprocess
    import sys
    r = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, timeout=120)
    assert r.returncode == 0 and r.stdout.strip() == "ok", r.stderr[-2000:]
