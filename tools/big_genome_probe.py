import sys, time
sys.path.insert(0, ".")
import torch, numpy as np
import hypergen_amd as hg
dev = torch.device("cuda:0")
ctx = hg.Context(0)
ctx.set_stream(torch.cuda.current_stream().cuda_stream)
p = hg.default_params()
for n, L in ((1, 500_000_000), (4, 100_000_000), (1, 3_000_000_000)):
    stride = (L + 1 + 15) // 16 * 16
    seq = torch.empty(n * stride + 64, dtype=torch.uint8, device=dev)
    ctx.synth_genomes_dev(0, n, L, stride, seq.data_ptr())
    offs = np.arange(n, dtype=np.uint64) * stride
    lens = np.full(n, L + 1, np.uint64)
    hv = torch.empty((n, 4096), dtype=torch.int16, device=dev)
    n2 = torch.empty(n, dtype=torch.int32, device=dev)
    nh = torch.empty(n, dtype=torch.int32, device=dev)
    ctx.enable_timing(True)
    for rep in range(2):
        ctx.timings()
        torch.cuda.synchronize(); t = time.time()
        ctx.sketch_batch_dev(seq.data_ptr(), offs, lens, p, hv.data_ptr(), n2.data_ptr(), nh.data_ptr())
        torch.cuda.synchronize(); dt = time.time() - t
        tm = ctx.timings()
    print("%d x %d bp: %.1f ms total; kernels %s; nhash %s; norm ok %s" % (
        n, L, dt * 1e3, {k: round(v[0], 2) for k, v in tm.items() if v[1]}, nh.tolist(),
        bool(((hv.int() ** 2).sum(1).int() == n2).all())))
    del seq
