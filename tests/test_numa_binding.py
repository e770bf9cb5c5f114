"""hg_bind_thread_to_numa_node (host code, no GPU): the CALLING thread is restricted to the CPUs of a NUMA node it may already
run on, never oversubscribing the node, and nothing else about the process changes."""
import os
import threading

import pytest

import hypergen_amd as hg


def node_cpus(node):
    path = "/sys/devices/system/node/node%d/cpulist" % node
    if not os.path.exists(path):
        return None
    cpus = set()
    for part in open(path).read().strip().split(","):
        a, _, b = part.partition("-")
        cpus |= set(range(int(a), int(b or a) + 1))
    return cpus


def in_thread(fn):
    out = {}

    def run():
        out["r"] = fn()
    t = threading.Thread(target=run)
    t.start()
    t.join()
    return out["r"]


def test_bind_restricts_only_the_calling_thread():
    cpus = node_cpus(0)
    if cpus is None:
        pytest.skip("no NUMA topology under /sys")
    before = os.sched_getaffinity(0)
    want = cpus & before
    if not want:
        pytest.skip("node 0 has no CPU this process may use")

    def bound():
        r = hg.lib().hg_bind_thread_to_numa_node(0, 1)
        return r, os.sched_getaffinity(threading.get_native_id())
    r, aff = in_thread(bound)
    assert r == 1 and aff == want
    assert os.sched_getaffinity(0) == before  # the main thread is where it was


def test_bind_refuses_to_oversubscribe_and_unknown_nodes():
    before = os.sched_getaffinity(0)

    def tries():
        lib = hg.lib()
        return (lib.hg_bind_thread_to_numa_node(0, 1 << 20), lib.hg_bind_thread_to_numa_node(-1, 1),
                lib.hg_bind_thread_to_numa_node(4095, 1), os.sched_getaffinity(threading.get_native_id()))
    a, b, c, aff = in_thread(tries)
    assert (a, b, c) == (0, 0, 0) and aff == before
