"""Multi-GPU decomposition of the hot path (one process per GPU, torch.distributed).

sketch: genomes are independent units (src/sketch.rs:35-56) -> contiguous index blocks per rank,
        NO data-path collective.
dist:   each rank keeps the query rows it owns and needs ALL reference HVs -> one all-gather of
        the R x D int16 reference matrix (+ R int32 norms) over RCCL/xGMI, then an independent
        R x (Q/world) block per rank.
The functions are backend-agnostic (nccl on MI355X, gloo in the CPU tests).
"""
import torch
import torch.distributed as dist


def shard_range(n, rank, world):
    """Contiguous block [lo, hi) of n units owned by `rank`; sizes differ by at most one."""
    base, rem = divmod(n, world)
    lo = rank * base + min(rank, rem)
    return lo, lo + base + (1 if rank < rem else 0)


def allgather_rows(local, world, group=None):
    """All-gather row blocks that may differ in size by one row (pads to the largest block)."""
    if world == 1:
        return local
    n_local = torch.tensor([local.shape[0]], dtype=torch.int64, device=local.device)
    sizes = [torch.zeros_like(n_local) for _ in range(world)]
    dist.all_gather(sizes, n_local, group=group)
    sizes = [int(s.item()) for s in sizes]
    mx = max(sizes)
    pad = local
    if local.shape[0] < mx:
        pad = torch.cat([local, local.new_zeros((mx - local.shape[0],) + tuple(local.shape[1:]))])
    # gather as raw bytes: the payload is opaque to the collective (and gloo has no int16)
    raw = pad.contiguous().view(torch.uint8)
    out = [torch.empty_like(raw) for _ in range(world)]
    dist.all_gather(out, raw, group=group)
    return torch.cat([o.view(local.dtype).reshape(pad.shape)[:s] for o, s in zip(out, sizes)])
