#!/usr/bin/env python3
"""Phase timing of the dist GEMM's main loop from in-kernel s_memtime stamps (development build:
tools/build_variant.sh stamps hg_dist_kernels.hip -DHG_DIST_STAMPS; run with HYPERGEN_LIB=tools/_exp_lib_stamps.so).
Prints, per wave kind (loader wave 0 / non-loading wave 5), the average length in shader cycles of the parts of a K-step."""
import ctypes as C
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import hypergen_amd as hg  # noqa: E402
import bench  # noqa: E402

n = int(os.environ.get("HG_N", 10000))
dev = torch.device("cuda:0")
ctx = hg.Context(0)
ctx.set_stream(torch.cuda.current_stream().cuda_stream)
hv = bench.clustered_hvs(n, 0, dev)
n2 = (hv.int() ** 2).sum(1).int()
cap = max(1 << 20, n * n // 20)
hits = torch.empty(cap * 3, dtype=torch.int32, device=dev)
for tile in os.environ.get("HG_TILES", "wide,big").split(","):
    ctx.set_debug("dist_tile", tile)
    for _ in range(30):
        ctx.dist_dev(hv.data_ptr(), n2.data_ptr(), n, hv.data_ptr(), n2.data_ptr(), n, 4096, 21, False, 85.0, hits.data_ptr(), cap)
    torch.cuda.synchronize()
    st = np.zeros((16, 2, 8, 6), np.uint64)
    assert hg.lib().hg_debug_dist_stamps(C.c_void_p(st.ctypes.data)) == 0
    st = st.astype(np.int64)
    names = ["top -> last fragments requested (7 phases of MFMAs)", "lgkmcnt wait", "barrier", "DMA issue", "last phase's MFMAs"]
    print("tile %s, kernel %s; s_memtime ticks (100 MHz constant clock on this part if the numbers look 20x too small)" % (tile, ctx.last_kernel("dist")))
    for w, wn in enumerate(("wave 0 (loader)", "wave 5 (no loads)")):
        d = np.diff(st[:, w], axis=2)  # [wg][step][5]
        step = st[:, w, 1:, 0] - st[:, w, :-1, 0]
        print("  %-18s K-step %.0f  |" % (wn, step.mean()), "  ".join("%s %.0f" % (nm, d[:, :, i].mean()) for i, nm in enumerate(names)))
