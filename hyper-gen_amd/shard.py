"""Multi-GPU decomposition of the hot path, one process per GPU (torch.distributed).

sketch: genomes are independent units (src/sketch.rs:35-56) -> contiguous index blocks per rank,
        NO data-path collective.
dist:   each rank keeps the query rows it owns and needs ALL reference HVs -> one all-gather of
        the R x D int16 reference matrix (+ R int32 norms) over RCCL/xGMI, then an independent
        R x (Q/world) block per rank.
search: the bit-packed reference database is sharded by rows, ONE query set is broadcast from
        rank 0, every rank searches its shard, the per-rank hit lists (global indices) are
        gathered on rank 0 (third row).
The functions are backend-agnostic (nccl = RCCL on MI355X, gloo in the CPU tests).  The same
partitioning inside ONE process is hg_multi in include/hypergen.h.
"""
import numpy as np
import torch
import torch.distributed as dist


def shard_range(n, rank, world):
    """Contiguous block [lo, hi) of n units owned by `rank`; sizes differ by at most one
    (same rule as hg_shard_range in the C ABI)."""
    base, rem = divmod(n, world)
    lo = rank * base + min(rank, rem)
    return lo, lo + base + (1 if rank < rem else 0)


def _sizes(n_local, world, device, group=None):
    t = torch.tensor([n_local], dtype=torch.int64, device=device)
    out = [torch.zeros_like(t) for _ in range(world)]
    dist.all_gather(out, t, group=group)
    return [int(s.item()) for s in out]


def allgather_rows(local, world, group=None):
    """All-gather row blocks that may differ in size by one row (pads to the largest block)."""
    if world == 1:
        return local
    sizes = _sizes(local.shape[0], world, local.device, group)
    mx = max(sizes)
    pad = local
    if local.shape[0] < mx:
        pad = torch.cat([local, local.new_zeros((mx - local.shape[0],) + tuple(local.shape[1:]))])
    # gather as raw bytes: the payload is opaque to the collective (RCCL and gloo have no int16)
    raw = pad.contiguous().view(torch.uint8)
    out = [torch.empty_like(raw) for _ in range(world)]
    dist.all_gather(out, raw, group=group)
    return torch.cat([o.view(local.dtype).reshape(pad.shape)[:s] for o, s in zip(out, sizes)])


def broadcast_rows(t, world, src=0, group=None, force=False):
    """Broadcast a tensor of any dtype from `src` as raw bytes (in place; returns t).  `force` runs the
    collective with one rank too (a one-GPU box then executes the RCCL code path)."""
    if world > 1 or force:
        dist.broadcast(t.view(torch.uint8).view(-1), src=src, group=group)
    return t


def gather_records(local, world, device=None, group=None, force=False):
    """Concatenate variable-length numpy record arrays (hit lists) of all ranks, in rank order.
    Every rank gets the result (an all-gather of padded byte blocks -- hit lists are small next to
    the operands); `device` is where the collective's tensors live (cuda for nccl, cpu for gloo)."""
    if world == 1 and not force:
        return local
    device = device or torch.device("cpu")
    raw = torch.from_numpy(np.ascontiguousarray(local).view(np.uint8).copy()).to(device)
    sizes = _sizes(raw.numel(), world, device, group)
    mx = max(max(sizes), 1)
    pad = torch.zeros(mx, dtype=torch.uint8, device=device)
    pad[: raw.numel()] = raw
    out = [torch.empty_like(pad) for _ in range(world)]
    dist.all_gather(out, pad, group=group)
    parts = [o[:s].cpu().numpy().view(local.dtype) for o, s in zip(out, sizes)]
    return np.concatenate(parts) if parts else local


def sharded_search(search_block, ref_local, ref_lo, queries, world, device=None, group=None, force=False):
    """Database search over a row-sharded reference set.

    ref_local : this rank's reference rows, global rows [ref_lo, ref_lo + len)
    queries   : the query set; rank 0's content is broadcast to every rank (in place)
    search_block(ref_local, ref_lo, queries) -> numpy record array of hits with GLOBAL ref_idx
    Returns the merged hit list (all ranks), in rank order.
    """
    broadcast_rows(queries, world, 0, group, force)
    hits = search_block(ref_local, ref_lo, queries)
    return gather_records(hits, world, device, group, force)
