#!/usr/bin/env python3
"""BASELINE configs[4] on one GPU: 50 000 ref x 10 000 query bit-packed D=16384 HVs, Hamming search; the three exact
paths side by side (HG_HAM_PATHS = comma list of hg_ctx_set_debug ham_path values; "" = the library's own choice)."""
import os
import sys

import torch

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import hypergen_amd as hg  # noqa: E402

R, Q, D = int(os.environ.get("HG_HAM_R", 50000)), int(os.environ.get("HG_HAM_Q", 10000)), 16384
dev = torch.device("cuda:0")
ctx = hg.Context(0)
ctx.set_stream(torch.cuda.current_stream().cuda_stream)
g = torch.Generator(device=dev)
g.manual_seed(5)
ref = torch.randint(-2**31, 2**31 - 1, (R, D // 32), dtype=torch.int32, device=dev, generator=g)
qry = ref[:Q].clone()
flip = torch.randint(0, 32, (Q, D // 32), device=dev, generator=g)
qry ^= (1 << flip.clamp(max=30)).int()  # one flipped bit per word: distance 512 to its own ref
cap = 1 << 22
hits = torch.empty(cap * 3, dtype=torch.int32, device=dev)
ctx.enable_timing(True)
ref_hits = None
for path in os.environ.get("HG_HAM_PATHS", "fp4,mfma,fp4,mfma,popc").split(","):
    ctx.set_debug("ham_path", path)
    for tile in os.environ.get("HG_HAM_TILES", "").split(","):
        ctx.set_debug("dist_tile", tile)
        best = None
        for rep in range(6):
            ctx.timings()
            n, st = ctx.hamming_search_dev(ref.data_ptr(), R, qry.data_ptr(), Q, D, 2000, hits.data_ptr(), cap)
            torch.cuda.synchronize()
            t = ctx.timings()
            ms, prep = t["dist"][0], t["dist_prep"][0]
            if rep >= 2:
                best = ms if best is None else min(best, ms)
        h = hits[: 3 * n].view(-1, 3).cpu()
        key = h[:, 0].long() * (1 << 32) + h[:, 1].long()
        h = h[torch.argsort(key)]
        if ref_hits is None:
            ref_hits = h
        same = ref_hits.shape == h.shape and bool((ref_hits == h).all())
        print("hamming %d x %d, D=%d, path %-5s tile %-5s -> code %d: %.3f ms (+ %.3f ms expansion) = %.1f k M pairs/s, %.2f PFLOP/s; hits %d, identical %s" % (
            R, Q, D, path or "auto", tile or "auto", ctx.last_hamming_path(), best, prep, R * Q / best / 1e6,
            2.0 * D * R * Q / best / 1e12, n, same), flush=True)
