#!/usr/bin/env python3
"""How fast does this box move 82 MB in + 82 MB out with a plain elementwise kernel? (development aid)"""
import torch
dev = torch.device("cuda:0")
for shape in ((10000, 4096), (40000, 4096)):
    a = torch.randint(-100, 100, shape, dtype=torch.int16, device=dev)
    b = torch.empty(shape, dtype=torch.float16, device=dev)
    for name, fn in (("copy_ i16->f16", lambda: b.copy_(a)), ("clone i16", lambda: a.clone())):
        ts = []
        for r in range(12):
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record(); fn(); e1.record(); torch.cuda.synchronize()
            ts.append(e0.elapsed_time(e1))
        ts.sort()
        mb = a.numel() * 2 * 2 / 1e6
        print("%s %s: median %.3f ms -> %.2f TB/s" % (name, shape, ts[len(ts) // 2], mb / ts[len(ts) // 2] / 1e6))
