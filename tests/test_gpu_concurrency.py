"""The inner seam under the reference's threading model: src/sketch_cuda.rs:79-96 calls the device function from up to
255 rayon workers that share device 0 (`gpu.bind_to_thread()` per worker, one synchronous call per file).  Here: 16 host
threads, each with its OWN hg_ctx on device 0, call hg_kmer_hash_sample (the literal replacement of
`extract_kmer_t1ha2_cuda`) and hg_sketch_batch on different genomes at the same time; every result must equal the
oracle's / the single-threaded one.  (ctypes releases the GIL for the duration of a foreign call, so the calls really
overlap.)"""
import threading

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

T = 16


@pytest.fixture(scope="module")
def hg():
    import hypergen_amd
    return hypergen_amd


def test_sixteen_threads_each_with_its_own_ctx_on_one_gpu(hg, orc):
    rng = np.random.default_rng(5)
    lens = [int(x) for x in rng.integers(60_000, 900_000, 3 * T)]
    genomes = [orc.synth_genome(g, L) for g, L in enumerate(lens)]
    genomes[5] = genomes[5].copy()
    genomes[5][2000:2100] = ord("N")
    genomes[7] = np.concatenate([genomes[7][:50_000], genomes[7][:50_000]])  # repeats: duplicates in the raw hits
    p = hg.default_params(scaled=200)
    with hg.Context(0) as one:
        want_hv, want_n2, want_nh = one.sketch_batch(genomes, p)
    want_sets = {g: orc.kmer_hash_sample(genomes[g], 21, 200) for g in range(0, len(genomes), 5)}
    for g in (0, 5, 7, 20):
        w_hv, w_n2, w_nh = orc.sketch_genome(genomes[g], scaled=200)
        assert want_nh[g] == w_nh and want_n2[g] == w_n2 and np.array_equal(want_hv[g], w_hv)
    ctxs = [hg.Context(0) for _ in range(T)]
    errors, start = [], threading.Barrier(T)
    got_hv = np.zeros_like(want_hv)
    got_n2 = np.zeros_like(want_n2)
    got_nh = np.zeros_like(want_nh)

    def worker(t):
        try:
            ctx = ctxs[t]
            start.wait()
            for rep in range(3):
                for g in range(t, len(genomes), T):  # one call per "file", like the reference's par_iter body
                    hs = ctx.kmer_hash_sample(genomes[g], 21, 200)
                    if g in want_sets and not np.array_equal(hs, want_sets[g]):
                        errors.append("hash set of genome %d (thread %d)" % (g, t))
                    if hs.size != want_nh[g]:
                        errors.append("hash count of genome %d" % g)
                    hv, n2, nh = ctx.sketch_batch([genomes[g]], p)
                    got_hv[g], got_n2[g], got_nh[g] = hv[0], n2[0], nh[0]
                # and a small batch per thread, interleaved with the others' single calls
                mine = list(range(t, len(genomes), T))
                hv, n2, nh = ctx.sketch_batch([genomes[g] for g in mine], p)
                if not (np.array_equal(hv, want_hv[mine]) and np.array_equal(n2, want_n2[mine]) and np.array_equal(nh, want_nh[mine])):
                    errors.append("batch of thread %d" % t)
        except Exception as e:  # pragma: no cover
            errors.append("thread %d: %r" % (t, e))

    th = [threading.Thread(target=worker, args=(t,)) for t in range(T)]
    for x in th:
        x.start()
    for x in th:
        x.join()
    for c in ctxs:
        c.close()
    assert not errors, errors[:5]
    assert np.array_equal(got_hv, want_hv) and np.array_equal(got_n2, want_n2) and np.array_equal(got_nh, want_nh)


def test_threads_sharing_one_ctx_is_serialised_by_the_caller(hg, orc):
    """hg_ctx is not re-entrant (its workspaces are per context, include/hypergen.h): the supported sharing pattern is one
    lock around each call -- the results must still be right when 8 threads take turns on ONE context."""
    genomes = [orc.synth_genome(g, 150_000 + 1000 * g) for g in range(16)]
    p = hg.default_params(scaled=100)
    lock = threading.Lock()
    out = [None] * len(genomes)
    with hg.Context(0) as ctx:
        def worker(t):
            for g in range(t, len(genomes), 8):
                with lock:
                    out[g] = ctx.sketch_batch([genomes[g]], p)
        th = [threading.Thread(target=worker, args=(t,)) for t in range(8)]
        for x in th:
            x.start()
        for x in th:
            x.join()
        hv, n2, nh = ctx.sketch_batch(genomes, p)
    for g in range(len(genomes)):
        assert np.array_equal(out[g][0][0], hv[g]) and out[g][1][0] == n2[g] and out[g][2][0] == nh[g]
