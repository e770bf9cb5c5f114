"""bench.py quotes profile counters only from summaries that belong to the tree and the kernel that ran (CPU test of the
stamp logic: no GPU, no torch)."""
import json
import os
import sys
import types

from conftest import ROOT


def test_profiled_requires_matching_stamp_and_kernel(tmp_path, monkeypatch):
    sys.path.insert(0, ROOT)
    import bench
    prof = tmp_path / "profiles"
    prof.mkdir()
    monkeypatch.setattr(bench, "ROOT", str(tmp_path))
    monkeypatch.setattr(bench, "hg", types.SimpleNamespace(source_stamp=lambda: "aaaa000011112222"))
    good = {"_stamp": {"source_sha": "aaaa000011112222", "kernels": ["kmer_sample_shared<21, true>"], "head": "c0ffee" * 6},
            "kmer_sample_shared<21, true>": {"hbm_bytes_per_launch": 5.0e9}}
    json.dump(good, open(prof / "r07_pmc.json", "w"))
    d, src = bench.profiled("_pmc.json", "kmer_sample_shared<21, true>")
    assert d is not None and "r07_pmc.json" in src and "c0ffeec0ffee" in src
    # another kernel ran than the one that was profiled
    d, src = bench.profiled("_pmc.json", "kmer_sample_shared<25, true>")
    assert d is None and "does not hold the kernel" in src
    # a newer summary taken on other sources shadows the good one: null, with the reason
    stale = dict(good, _stamp=dict(good["_stamp"], source_sha="ffff000011112222"))
    json.dump(stale, open(prof / "r08_pmc.json", "w"))
    d, src = bench.profiled("_pmc.json", "kmer_sample_shared<21, true>")
    assert d is None and "other sources" in src
    # no stamp at all (the pre-round-3 files)
    json.dump({"kmer_sample_grouped<21>": {}}, open(prof / "r09_pmc.json", "w"))
    assert bench.profiled("_pmc.json", "kmer_sample_grouped<21>")[0] is None
    assert bench.profiled("_nothing.json", "x")[0] is None


def test_committed_profiles_carry_stamps():
    import glob
    files = sorted(glob.glob(os.path.join(ROOT, "profiles", "r03_*.json")))
    assert files
    for f in files:
        if f.endswith(("_bench_line.json",)):
            continue
        d = json.load(open(f))
        sha = (d.get("_stamp") or {}).get("source_sha") or d.get("source_sha")
        assert sha and len(sha) == 16, f
