"""bench.py quotes profile counters only from summaries that belong to the tree and the kernel that ran (CPU test of the
stamp logic: no GPU, no torch)."""
import json

import numpy as np
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


def test_draftify_makes_what_it_says():
    """bench.draftify (the `realistic.draft_assemblies` input): 200 contigs, ~30 % soft-masked in 1 kbp runs, 0.05 % IUPAC codes,
    one tandem repeat of a 171-base unit -- checked on the CPU on three small genomes (the generator runs on whatever device
    the buffer lives on)"""
    import torch
    import bench
    n, L = 3, 300_000
    stride = (L + 1 + 15) // 16 * 16
    g = torch.Generator().manual_seed(1)
    seq = torch.tensor(list(b"ACGT"), dtype=torch.uint8)[torch.randint(0, 4, (n * stride + 64,), generator=g)]
    seq.view(-1)[0: n * stride: stride] = ord("N")
    clean = seq.clone()
    bench.draftify(seq, n, stride, L, repeat_len=50_000)
    a = seq[: n * stride].view(n, stride).numpy()
    c = clean[: n * stride].view(n, stride).numpy()
    assert (a[:, 0] == ord("N")).all() and np.array_equal(a[:, L + 1:], c[:, L + 1:])  # nothing outside the bodies moved
    body = a[:, 1: 1 + L]
    for row in body:
        assert 150 <= (row == ord("N")).sum() <= 199            # the record starts (a few coincide / are overwritten)
        low = np.isin(row, np.frombuffer(b"acgt", np.uint8)).mean()
        assert 0.2 < low < 0.4                                    # soft-masked share
        iupac = np.isin(row, np.frombuffer(b"RYKMSWBDHVn", np.uint8)).sum()
        assert 0.0003 * L < iupac <= 0.0005 * L + 1
        up = row & 0xDF
        # the tandem repeat: somewhere a 50 kbp stretch in which base i equals base i + 171 (but for the sprinkled codes)
        same = (up[:-171] == up[171:]).astype(np.int32)
        cs = np.concatenate([[0], np.cumsum(same)])
        run = cs[40_000:] - cs[:-40_000]  # matches in every window of 40 000 positions
        assert run.max() > 0.995 * 40_000
