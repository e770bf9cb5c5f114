#!/usr/bin/env python3
"""List-scheduling model of one XCD of a dist launch (32 CUs, workgroup slots handed out in order to whichever CU is free):
what the launch's length is for a given mix of tile durations -- the arithmetic behind DESIGN.md 4.3's "5.3 tile times of
work per CU take 6.2".  Durations in k ticks from tools/dist_cu_timeline.py (10 000 x 10 000, 1.29 M hits, round 4):
normal tile 135, first diagonal tile (dense) 246, second diagonal tile 186; without candidates every tile 117.
usage: tools/dist_schedule_sim.py [dense second normal]"""
import heapq
import random
import sys


def makespan(jobs, cus=32):
    h = [0.0] * cus
    heapq.heapify(h)
    end = 0.0
    for j in jobs:
        t = heapq.heappop(h) + j
        end = max(end, t)
        heapq.heappush(h, t)
    return end


def jit(x, s):
    return x * (1 + random.gauss(0, s))


def run(dense, second, normal, label):
    random.seed(1)
    res = []
    for _ in range(400):
        jobs = [jit(dense, 0.08) for _ in range(5)] + [jit(second, 0.08) for _ in range(5)] + [jit(normal, 0.03) for _ in range(150)]
        res.append(makespan(jobs))
    work = (5 * dense + 5 * second + 150 * normal) / 32.0
    print("%-44s work per CU %5.0f k = %.2f tiles, launch %5.0f k = %.2f tiles" % (label, work, work / normal, sum(res) / len(res), sum(res) / len(res) / normal))


if len(sys.argv) == 4:
    run(*[float(x) for x in sys.argv[1:]], "as given")
else:
    run(117, 117, 117, "no candidates")
    run(308, 233, 134, "round 3 epilogue (slab path for dense tiles)")
    run(246, 186, 135, "round 4 (dense tiles in groups of slabs)")
    run(203, 183, 124.5, "deferred evaluation (tile kernel alone)")
    run(180, 150, 135, "if dense tiles cost 180 / 150")
    run(135, 135, 135, "if no tile stood out")
