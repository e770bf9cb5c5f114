"""The sync-free sketch step (hyper-gen_amd/csrc/hg_sketch_step.hip): hg_sketch_batch_dev queues hash + sample -> sort /
unique -> encode without a host round trip and reads ONE check word a call late.  What the host used to decide between
the kernels (src/sketch.rs:35-56 has no such seam: one task per file) must come out identical:
  * same sketches as the synchronous path and as the oracle, both input forms, ragged batches, repeated geometry;
  * a genome that overflows its hit region, or outgrows the one-workgroup sort, is marked HG_NHASH_PENDING in stream order
    and is final after hg_ctx_sync / the next call on the ctx -- also when two steps share their output buffers.
"""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

ACGT = np.frombuffer(b"ACGT", np.uint8)
PENDING = 0xFFFFFFFF


@pytest.fixture(scope="module")
def hg():
    import hypergen_amd
    return hypergen_amd


@pytest.fixture(params=["ascii", "packed"])
def ctx(hg, request):
    c = hg.Context(0)
    c.set_debug("kmer_input", "packed" if request.param == "packed" else "")
    yield c
    c.close()


def upload(seqs):
    offs, total = [], 0
    for s in seqs:
        offs.append(total)
        total += (len(s) + 15) // 16 * 16
    host = np.zeros(total + 64, np.uint8)
    for o, s in zip(offs, seqs):
        host[o:o + len(s)] = s
    return torch.from_numpy(host).cuda(), np.array(offs, np.uint64), np.array([len(s) for s in seqs], np.uint64)


def outputs(n, hv_d=4096):
    dev = torch.device("cuda:0")
    return (torch.full((n, hv_d), 7, dtype=torch.int16, device=dev), torch.full((n,), 7, dtype=torch.int32, device=dev),
            torch.full((n,), 7, dtype=torch.int32, device=dev))


def sampled_kmer(orc, rng, scaled):
    thr = (2**64 - 1) // scaled
    for _ in range(200000):
        cand = rng.choice(ACGT, 21)
        if orc.kmer_hash_sample(cand, 21, threshold=thr).size == 1:
            return cand
    raise AssertionError("no sampled k-mer found")


def repeat_genome(orc, rng, scaled, repeats, tail):
    unit = np.concatenate([sampled_kmer(orc, rng, scaled), np.frombuffer(b"N", np.uint8)])
    return np.concatenate([np.tile(unit, repeats), rng.choice(ACGT, tail)])


def check(orc, seqs, hv, n2, nh, **kw):
    hv, n2, nh = hv.cpu().numpy(), n2.cpu().numpy(), nh.cpu().numpy().view(np.uint32)
    for i, s in enumerate(seqs):
        w = orc.sketch_genome(s, **kw)
        assert nh[i] == w[2] and n2[i] == w[1] and (hv[i] == w[0]).all(), i


def test_sync_free_equals_synchronous_path_and_oracle(hg, ctx, orc):
    rng = np.random.default_rng(601)
    lens = [0, 5, 21, 300, 3048, 3049, 27432, 27433, 60_000, 150_001, 400_000, 1_000_003]
    seqs = [rng.choice(ACGT, n) for n in lens]
    seqs[8][1000:1200] = ord("N")
    d_seq, offs, ln = upload(seqs)
    p = hg.default_params()
    hv, n2, nh = outputs(len(seqs))
    f0, s0, r0 = ctx.sketch_step_counts()
    ctx.sketch_batch_dev(d_seq.data_ptr(), offs, ln, p, hv.data_ptr(), n2.data_ptr(), nh.data_ptr())
    ctx.sync()
    f1, s1, r1 = ctx.sketch_step_counts()
    assert (f1 - f0, s1 - s0, r1 - r0) == (1, 0, 0)
    check(orc, seqs, hv, n2, nh)
    hv2, n22, nh2 = outputs(len(seqs))
    ctx.set_debug("sketch_path", "sync")
    ctx.sketch_batch_dev(d_seq.data_ptr(), offs, ln, p, hv2.data_ptr(), n22.data_ptr(), nh2.data_ptr())
    ctx.sync()
    ctx.set_debug("sketch_path", "")
    f2, s2, r2 = ctx.sketch_step_counts()
    assert (f2 - f1, s2 - s1, r2 - r1) == (0, 1, 0)
    assert torch.equal(hv, hv2) and torch.equal(n2, n22) and torch.equal(nh, nh2)


def test_repeated_and_changing_geometry_without_a_sync_in_between(hg, ctx, orc):
    """a stream of steps, each read only after the whole stream: cached plan, new plan, cached again, another k"""
    rng = np.random.default_rng(602)
    a = [rng.choice(ACGT, n) for n in (50_000, 80_000, 120_000)]
    b = [rng.choice(ACGT, n) for n in (70_000, 10, 90_000, 200_000)]
    da, oa, la = upload(a)
    db, ob, lb = upload(b)
    p = hg.default_params()
    p25 = hg.default_params(ksize=25, scaled=300)
    outs = []
    for d, o, l, seqs, pp in ((da, oa, la, a, p), (da, oa, la, a, p), (db, ob, lb, b, p), (da, oa, la, a, p), (db, ob, lb, b, p25),
                              (db, ob, lb, b, p25)):
        hv, n2, nh = outputs(len(seqs))
        ctx.sketch_batch_dev(d.data_ptr(), o, l, pp, hv.data_ptr(), n2.data_ptr(), nh.data_ptr())
        outs.append((seqs, pp, hv, n2, nh))
    ctx.sync()
    assert ctx.sketch_step_counts()[2] == 0
    for seqs, pp, hv, n2, nh in outs:
        check(orc, seqs, hv, n2, nh, ksize=pp.ksize, scaled=pp.scaled)


def test_overflowing_genome_is_pending_in_stream_order_and_final_after_sync(hg, ctx, orc):
    rng = np.random.default_rng(603)
    bad = repeat_genome(orc, rng, 1500, 4000, 100_000)  # 4 000 raw hits, region of ~1.2 k
    seqs = [orc.synth_genome(5, 150_000), bad, orc.synth_genome(6, 90_000)]
    d_seq, offs, ln = upload(seqs)
    p = hg.default_params()
    hv, n2, nh = outputs(3)
    ctx.sketch_batch_dev(d_seq.data_ptr(), offs, ln, p, hv.data_ptr(), n2.data_ptr(), nh.data_ptr())
    torch.cuda.synchronize()  # the stream alone: NOT the library's completion point
    got = nh.cpu().numpy().view(np.uint32)
    assert got[1] == PENDING and (hv[1] == 7).all()  # marked, row untouched
    for i in (0, 2):  # the other genomes of the step are final
        w = orc.sketch_genome(seqs[i])
        assert got[i] == w[2] and (hv[i].cpu().numpy() == w[0]).all()
    before = ctx.sketch_step_counts()
    ctx.sync()
    after = ctx.sketch_step_counts()
    assert after[2] - before[2] == 1 and after[1] - before[1] == 1  # run again, through the synchronous path
    check(orc, seqs, hv, n2, nh)


def test_set_beyond_the_one_workgroup_sort_is_redone(hg, ctx, orc):
    """expected 5 880 sampled k-mers (scaled = 100), region 12 782: 4 000 repeats of a sampled k-mer make ~9 000 raw hits --
    inside the region, beyond the 8 192 keys one workgroup sorts"""
    rng = np.random.default_rng(604)
    big = repeat_genome(orc, rng, 100, 4000, 500_000)
    seqs = [big, rng.choice(ACGT, 30_000)]
    d_seq, offs, ln = upload(seqs)
    p = hg.default_params(scaled=100)
    hv, n2, nh = outputs(2)
    ctx.sketch_batch_dev(d_seq.data_ptr(), offs, ln, p, hv.data_ptr(), n2.data_ptr(), nh.data_ptr())
    torch.cuda.synchronize()
    assert nh.cpu().numpy().view(np.uint32)[0] == PENDING
    before = ctx.sketch_step_counts()
    ctx.sync()
    assert ctx.sketch_step_counts()[2] - before[2] == 1
    check(orc, seqs, hv, n2, nh, scaled=100)


def test_two_steps_sharing_output_buffers_keep_call_order(hg, ctx, orc):
    """step 1 overflows and is read only while step 2 is being queued: its re-run lands BEHIND step 2 on the stream, so
    step 2 is run again as well and the shared buffers hold step 2's sketches"""
    rng = np.random.default_rng(605)
    first = [repeat_genome(orc, rng, 1500, 3000, 60_000), rng.choice(ACGT, 60_000)]
    second = [rng.choice(ACGT, 3000 * 22 + 60_000), rng.choice(ACGT, 60_000)]  # same geometry, other content
    assert [len(s) for s in first] == [len(s) for s in second]
    d1, offs, ln = upload(first)
    d2, _, _ = upload(second)
    p = hg.default_params()
    hv, n2, nh = outputs(2)
    ctx.sketch_batch_dev(d1.data_ptr(), offs, ln, p, hv.data_ptr(), n2.data_ptr(), nh.data_ptr())
    ctx.sketch_batch_dev(d2.data_ptr(), offs, ln, p, hv.data_ptr(), n2.data_ptr(), nh.data_ptr())
    ctx.sync()
    check(orc, second, hv, n2, nh)
    assert ctx.sketch_step_counts()[2] >= 1


def test_any_later_call_on_the_ctx_completes_the_step(hg, ctx, orc):
    """no hg_ctx_sync: hg_copy_d2h (an ordinary entry point) reads the check word first"""
    rng = np.random.default_rng(606)
    seqs = [repeat_genome(orc, rng, 1500, 2500, 40_000), rng.choice(ACGT, 33_333)]
    d_seq, offs, ln = upload(seqs)
    p = hg.default_params()
    hv, n2, nh = outputs(2)
    ctx.sketch_batch_dev(d_seq.data_ptr(), offs, ln, p, hv.data_ptr(), n2.data_ptr(), nh.data_ptr())
    got = np.zeros(2, np.uint32)
    import ctypes
    ctx._ck(hg.lib().hg_copy_d2h(ctx._h, ctypes.c_void_p(got.ctypes.data), ctypes.c_void_p(nh.data_ptr()), 8))
    assert got[0] == orc.sketch_genome(seqs[0])[2] and got[1] == orc.sketch_genome(seqs[1])[2]
    check(orc, seqs, hv, n2, nh)


def test_count_sized_sort_leaves_no_genome_behind(hg, ctx, orc):
    """the first sort launch is sized for 1.125 x the expected count; a genome with 1.5 x as many distinct hashes as
    expected (a sampled stretch repeated with a one-base change) is taken by the second launch on the device -- no re-run"""
    rng = np.random.default_rng(607)
    base = rng.choice(ACGT, 600_000)
    extra = []
    while len(extra) < 300:  # 300 more DISTINCT sampled k-mers on top of the expected 400
        cand = rng.choice(ACGT, 21)
        if orc.kmer_hash_sample(cand, 21, 1500).size == 1:
            extra.append(np.concatenate([cand, np.frombuffer(b"N", np.uint8)]))
    heavy = np.concatenate(extra + [base])
    seqs = [heavy, rng.choice(ACGT, 600_000)]
    d_seq, offs, ln = upload(seqs)
    p = hg.default_params()
    hv, n2, nh = outputs(2)
    ctx.sketch_batch_dev(d_seq.data_ptr(), offs, ln, p, hv.data_ptr(), n2.data_ptr(), nh.data_ptr())
    before = ctx.sketch_step_counts()
    ctx.sync()
    assert ctx.sketch_step_counts()[2] == before[2]
    check(orc, seqs, hv, n2, nh)
    assert nh.cpu().numpy().view(np.uint32)[0] > 400 * 1.125 + 64


def test_stream_of_files_with_an_overflowing_genome(hg, orc):
    """the streaming entry points consume their chunks in stream order: the chunk with the overflowing genome is fetched again"""
    rng = np.random.default_rng(608)
    seqs = [rng.choice(ACGT, 50_000), repeat_genome(orc, rng, 1500, 3500, 70_000), rng.choice(ACGT, 20_000)]
    st = hg.SketchStream([0], hg.default_params())
    for i, s in enumerate(seqs):
        st.push(s, i)
    st.finish()
    seen = {}
    while True:
        r = st.pop()
        if r is None:
            break
        seen[r[0]] = r[1:]
    st.close()
    assert sorted(seen) == [0, 1, 2]
    for i, s in enumerate(seqs):
        w = orc.sketch_genome(s)
        hv, n2, nh = seen[i]
        assert nh == w[2] and n2 == w[1] and (hv == w[0]).all(), i


@pytest.mark.parametrize("seed", range(4))
def test_random_sequences_of_steps(hg, ctx, orc, seed):
    """a soak of the check-word logic: forty steps of random batches -- repeated geometry, new geometry, overflowing genomes one
    step in five, output buffers drawn from a pool of two (so a re-run may land behind a later step that shares them), other
    entry points in between -- read only at the end: every buffer must hold the sketches of the LAST step that wrote it"""
    rng = np.random.default_rng(700 + seed)
    p = hg.default_params(scaled=int(rng.choice([300, 1500])))
    pool = [outputs(12), outputs(12)]
    last = [None, None]  # per buffer: the sequences of the last step that wrote it
    keep = []            # device inputs stay alive until the final sync (the contract)
    prev = None
    kmer = sampled_kmer(orc, rng, int(p.scaled))
    for step in range(40):
        if prev is not None and rng.random() < 0.3:
            seqs = [rng.choice(ACGT, len(s)) for s in prev]  # same geometry, other content
        else:
            seqs = [rng.choice(ACGT, int(n)) for n in rng.choice([0, 20, 500, 3048, 9000, 27500, 60_000, 150_000], int(rng.integers(1, 12)))]
        if rng.random() < 0.2 and len(seqs[0]) >= 60_000:  # a sampled k-mer 3 000 times: overflows a region of ~1 100
            unit = np.concatenate([kmer, np.frombuffer(b"N", np.uint8)])
            seqs[0] = np.concatenate([np.tile(unit, 3000), seqs[0][3000 * 22:]])[:len(seqs[0])]
        prev = seqs
        d_seq, offs, ln = upload(seqs)
        keep.append(d_seq)
        b = int(rng.integers(0, 2))
        hv, n2, nh = pool[b]
        ctx.sketch_batch_dev(d_seq.data_ptr(), offs, ln, p, hv.data_ptr(), n2.data_ptr(), nh.data_ptr())
        last[b] = seqs
        if rng.random() < 0.15:  # any other entry point completes the queued step first
            import ctypes
            tmp = np.zeros(1, np.uint32)
            ctx._ck(hg.lib().hg_copy_d2h(ctx._h, ctypes.c_void_p(tmp.ctypes.data), ctypes.c_void_p(nh.data_ptr()), 4))
    ctx.sync()
    fast, slow, rerun = ctx.sketch_step_counts()
    assert fast >= 30
    for b in range(2):
        if last[b] is None:
            continue
        hv, n2, nh = pool[b]
        k = len(last[b])
        check(orc, last[b], hv[:k], n2[:k], nh[:k], scaled=int(p.scaled))
