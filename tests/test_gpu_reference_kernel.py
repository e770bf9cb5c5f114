"""The reference's OWN kernel (src/cuda_kernel.cu built in place by hipcc into oracle/_ref/) run on
the MI355X next to the oracle and the product: three-way agreement on the sampled hash sets.
k <= 24 with the -O3 build of the reference source, k = 25..32 (the `default:` branch of its t1ha2_atonce: the
prime_4 stage) with the -O0 build of the same source -- optimised builds of the reference kernel fault there."""
import os
import subprocess
import tempfile

import numpy as np
import pytest

from conftest import ROOT

pytestmark = pytest.mark.gpu
REF = os.path.join(ROOT, "oracle", "_ref")
HAVE_REF = all(os.path.exists(os.path.join(REF, f)) for f in ("ref_kmer_runner", "ref_cuda_kernel.hsaco", "ref_cuda_kernel_O0.hsaco"))


def run_ref(seq, k, scaled, seed=123, canonical=True, slots=0):
    with tempfile.TemporaryDirectory() as td:
        fi, fo = os.path.join(td, "s.bin"), os.path.join(td, "o.bin")
        np.ascontiguousarray(seq, np.uint8).tofile(fi)
        hsaco = "ref_cuda_kernel.hsaco" if k <= 24 else "ref_cuda_kernel_O0.hsaco"
        subprocess.check_call([os.path.join(REF, "ref_kmer_runner"), os.path.join(REF, hsaco),
                               fi, str(k), str(scaled), str(seed), "1" if canonical else "0", str(slots), fo])
        return np.fromfile(fo, np.uint64)


@pytest.mark.skipif(not HAVE_REF, reason="oracle/_ref not built (needs /root/reference at build time)")
@pytest.mark.parametrize("g,L,k,scaled,canon,slots", [
    (11, 150_000, 21, 1500, True, 0),     # the reference's own launch parameters (8 slots/thread)
    (12, 150_000, 21, 1, True, 520),      # every k-mer
    (13, 80_000, 21, 20, False, 128),     # non-canonical mode (src/cuda_kernel.cu:312-314)
    (14, 80_000, 12, 20, True, 128),
    (15, 80_000, 24, 20, True, 128),
    (16, 80_000, 16, 3, True, 256),
    (17, 60_000, 25, 20, True, 128),      # k >= 25: reference kernel built -O0
    (18, 60_000, 27, 5, False, 256),
    (19, 60_000, 31, 20, True, 128),
    (20, 60_000, 32, 1, True, 520),
])
def test_three_way_hash_sets(orc, g, L, k, scaled, canon, slots):
    import hypergen_amd as hg
    seq = orc.synth_genome(g, L).copy()
    rng = np.random.default_rng(g)
    seq[rng.choice(L, 30, replace=False) + 1] = ord("N")
    seq[1000:1500] = np.char.lower(seq[1000:1500].view("S1")).view(np.uint8)
    ref = run_ref(seq, k, scaled, canonical=canon, slots=slots)
    want = orc.kmer_hash_sample(seq, k, scaled, 123, canon)
    with hg.Context(0) as ctx:
        got = ctx.kmer_hash_sample(seq, k, scaled, 123, canon)
    assert ref.size == want.size and (ref == want).all(), "reference kernel vs oracle"
    assert got.size == want.size and (got == want).all(), "product vs oracle"


@pytest.mark.skipif(not HAVE_REF, reason="oracle/_ref not built (needs /root/reference at build time)")
@pytest.mark.parametrize("case", range(60))
def test_three_way_random_cases(orc, case):
    """Random k, sampling rate, seed, strand mode and contamination (N runs, lower case, IUPAC and other bytes, record
    separators at random places), k = 1..32: the reference's own kernel, the oracle and the product must give the same hash set."""
    import hypergen_amd as hg
    rng = np.random.default_rng(31_000 + case)
    L = int(rng.integers(600, 40_000))
    k = int(rng.integers(1, 33))
    scaled = int(rng.choice([1, 2, 3, 7, 20, 50, 200]))
    seed = int(rng.integers(0, 2**63))
    canon = bool(rng.integers(0, 2))
    seq = orc.synth_genome(500 + case, L).copy()
    for _ in range(int(rng.integers(0, 6))):  # runs of not-a-base bytes
        a = int(rng.integers(1, L))
        seq[a:a + int(rng.integers(1, 40))] = int(rng.choice(list(b"NnRYKMSWryX-*.")))
    for _ in range(int(rng.integers(0, 4))):  # soft-masked stretches
        a = int(rng.integers(1, L))
        b = min(L + 1, a + int(rng.integers(1, 3000)))
        seq[a:b] = np.char.lower(seq[a:b].view("S1")).view(np.uint8)
    seq[rng.choice(L, int(rng.integers(0, 20)), replace=False) + 1] = ord("N")  # record separators / single N's
    ref = run_ref(seq, k, scaled, seed=seed, canonical=canon, slots=520)
    want = orc.kmer_hash_sample(seq, k, scaled, seed, canon)
    with hg.Context(0) as ctx:
        for form in ("ascii", "packed"):
            ctx.set_debug("kmer_input", form)
            got = ctx.kmer_hash_sample(seq, k, scaled, seed, canon)
            assert got.size == want.size and (got == want).all(), ("product vs oracle", form, k, scaled, canon, L)
    assert ref.size == want.size and (ref == want).all(), ("reference kernel vs oracle", k, scaled, canon, L)


def _g2_cases():
    from conftest import golden
    return golden("g2_ref_kernel.json")


@pytest.mark.parametrize("idx", range(13))
def test_g2_reference_fixture_through_the_product(orc, idx):
    """The hash sets the reference's own kernel (src/cuda_kernel.cu:250-321, built in place by hipcc,
    tools/gen_golden_ref_gpu.py) produced on the MI355X, committed as tests/golden/g2_ref_kernel.json, against
    hg_kmer_hash_sample -- size, xor, sum and the stored hashes.  Needs nothing from oracle/_ref/ on the box: the
    fixture is what survives without the reference checkout.  (orc only regenerates the seeded INPUT.)"""
    import hypergen_amd as hg
    cases = _g2_cases()
    assert len(cases) == 13
    c = cases[idx]
    seq = orc.synth_genome(c["genome"], c["L"])
    if c["mutated"]:  # the generator's edits (tools/gen_golden_ref_gpu.py): 50 N's and a lower-case stretch
        rng = np.random.default_rng(c["genome"])
        seq = seq.copy()
        seq[rng.choice(c["L"], 50, replace=False) + 1] = ord("N")
        seq[5000:6000] = np.char.lower(seq[5000:6000].view("S1")).view(np.uint8)
    with hg.Context(0) as ctx:
        for form in ("ascii", "packed"):
            ctx.set_debug("kmer_input", form)
            got = ctx.kmer_hash_sample(seq, c["k"], c["scaled"], 123, c["canonical"])
            assert got.size == c["n"], (c["name"], form)
            assert "%016x" % int(np.bitwise_xor.reduce(got) if got.size else 0) == c["xor"], (c["name"], form)
            assert "%016x" % (int(got.astype(object).sum()) % 2**64 if got.size else 0) == c["sum"], (c["name"], form)
            want = np.array([int(h, 16) for h in c["hashes"]], np.uint64)
            if c["subsampled"]:
                assert np.isin(want, got).all(), (c["name"], form)
            else:
                assert (want == got).all(), (c["name"], form)
