"""2-bit packed bases as the RESIDENT input form of the sketch path (north_star: "coalesced HBM reads of packed bases";
the reference's second kernel works on NT4 codes, src/cuda_kernel.cu:15-69, src/sketch_cuda.rs:23-32).
hg_sketch_batch_dev_packed reads hg_pack2 blobs as they lie -- no ASCII copy exists on the device -- and must give the
hash sets / HVs / norms of hg_sketch_batch_dev on the sequences the blobs were packed from, and of the CPU oracle.
hg_pack2_batch_dev (the device packer) must reproduce the host's hg_pack2 byte for byte."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

ALPHA = np.frombuffer(b"ACGTacgtNnRYKMUu-.*", np.uint8)


@pytest.fixture(scope="module")
def hg():
    import hypergen_amd
    return hypergen_amd


@pytest.fixture(scope="module")
def ctx(hg):
    c = hg.Context(0)
    c.set_stream(torch.cuda.current_stream().cuda_stream)
    yield c
    c.close()


def genome(rng, n, junk=0.002):
    s = rng.choice(ALPHA[:4], n)
    if n:
        s = np.where(rng.random(n) < 0.05, s | 0x20, s).astype(np.uint8)
        s = np.where(rng.random(n) < junk, rng.choice(ALPHA[4:], n), s).astype(np.uint8)
        if n > 600:
            a = int(rng.integers(0, n - 300))
            s[a:a + int(rng.integers(1, 300))] = ord("N")
    return s


def upload(seqs, align):
    """genomes back to back at `align`-byte aligned offsets + 64 bytes of slack; returns (device tensor, offsets, lens)"""
    offs, total = [], 0
    for s in seqs:
        offs.append(total)
        total += (len(s) + align - 1) // align * align
    host = np.zeros(total + 64, np.uint8)
    for o, s in zip(offs, seqs):
        host[o:o + len(s)] = s
    return torch.from_numpy(host).cuda(), np.array(offs, np.uint64), np.array([len(s) for s in seqs], np.uint64)


LENGTHS = [0, 1, 3, 20, 21, 22, 31, 32, 33, 63, 64, 65, 127, 128, 129, 255, 257, 511, 1000, 3047, 3048, 3049, 3071, 3072,
           3073, 6096, 8191, 8192, 8193, 27431, 27432, 27433, 100_003]


@pytest.mark.parametrize("norm", [0, 1])
def test_device_packer_equals_host_pack2(hg, ctx, norm):
    rng = np.random.default_rng(40 + norm)
    seqs = [genome(rng, n, junk=0.02) for n in LENGTHS]
    d_seq, offs, lens = upload(seqs, 4)
    sizes = [hg.lib().hg_pack2_size(len(s)) for s in seqs]
    boffs = np.concatenate([[0], np.cumsum(sizes)[:-1]]).astype(np.uint64)
    d_blobs = torch.full((int(sum(sizes)) + 64,), 0xEE, dtype=torch.uint8, device="cuda")
    ctx.pack2_batch_dev(d_seq.data_ptr(), offs, lens, d_blobs.data_ptr(), boffs, norm)
    got = d_blobs.cpu().numpy()
    for s, o, z in zip(seqs, boffs, sizes):
        assert np.array_equal(got[int(o):int(o) + z], hg.pack2(s, norm)), len(s)
    assert (got[int(sum(sizes)):] == 0xEE).all()  # nothing written behind the last blob


@pytest.mark.parametrize("k,scaled,canon", [(21, 50, True), (21, 1, True), (12, 20, False), (16, 30, True), (25, 40, True),
                                            (32, 25, True), (31, 25, False), (33, 40, True), (64, 40, False), (5, 400, True)])
def test_packed_batch_equals_ascii_batch_and_oracle(hg, ctx, orc, k, scaled, canon):
    rng = np.random.default_rng(1000 + k)
    lens = LENGTHS if k == 21 and scaled == 50 else [0, 5, k - 1, k, k + 1, 100, 3050, 9000, 27433, 60_001]
    seqs = [genome(rng, n) for n in lens]
    p = hg.default_params(ksize=k, scaled=scaled, canonical=int(canon), hv_d=1024)
    n = len(seqs)
    blobs = [hg.pack2(s) for s in seqs]
    d_blobs, boffs, _ = upload(blobs, 16)
    d_seq, offs, ln = upload(seqs, 16)
    out = []
    for form in ("ascii", "packed", "device-packed"):
        hv = torch.zeros((n, 1024), dtype=torch.int16, device="cuda")
        n2 = torch.zeros(n, dtype=torch.int32, device="cuda")
        nh = torch.zeros(n, dtype=torch.int32, device="cuda")
        if form == "ascii":
            ctx.sketch_batch_dev(d_seq.data_ptr(), offs, ln, p, hv.data_ptr(), n2.data_ptr(), nh.data_ptr())
        elif form == "packed":
            ctx.sketch_batch_dev_packed(d_blobs.data_ptr(), boffs, ln, p, hv.data_ptr(), n2.data_ptr(), nh.data_ptr())
            assert ctx.last_kernel("kmer").endswith("true>")  # kmer_sample_shared<k, canon, true> / kmer_sample_long<true>
        else:
            d2 = torch.zeros_like(d_blobs)
            ctx.pack2_batch_dev(d_seq.data_ptr(), offs, ln, d2.data_ptr(), boffs)
            ctx.sketch_batch_dev_packed(d2.data_ptr(), boffs, ln, p, hv.data_ptr(), n2.data_ptr(), nh.data_ptr())
        ctx.sync()  # (reads the step's check word, then waits for the stream)
        out.append((hv.cpu().numpy(), n2.cpu().numpy(), nh.cpu().numpy()))
    for o in out[1:]:
        assert np.array_equal(o[0], out[0][0]) and np.array_equal(o[1], out[0][1]) and np.array_equal(o[2], out[0][2])
    for i in (1, 3, 4, 5, len(seqs) - 2, len(seqs) - 1):
        w_hv, w_n2, w_nh = orc.sketch_genome(seqs[i], k, scaled, 123, canon, hv_d=1024)
        assert out[1][2][i] == w_nh and out[1][1][i] == w_n2 and np.array_equal(out[1][0][i], w_hv), (i, len(seqs[i]))


def test_packed_u2t_blobs(hg, ctx, orc):
    """a blob packed under HG_NORM_U2T carries u/U as T: the packed kernels give the U2T result whatever p.norm_mode says"""
    rng = np.random.default_rng(7)
    s = genome(rng, 50_000, junk=0.0)
    s[rng.choice(50_000, 500, replace=False)] = ord("U")
    s[rng.choice(50_000, 500, replace=False)] = ord("u")
    for norm in (0, 1):
        blob = torch.from_numpy(np.concatenate([hg.pack2(s, norm), np.zeros(64, np.uint8)])).cuda()
        p = hg.default_params(scaled=20, hv_d=512)
        hv = torch.zeros((1, 512), dtype=torch.int16, device="cuda")
        n2 = torch.zeros(1, dtype=torch.int32, device="cuda")
        nh = torch.zeros(1, dtype=torch.int32, device="cuda")
        ctx.sketch_batch_dev_packed(blob.data_ptr(), [0], [s.size], p, hv.data_ptr(), n2.data_ptr(), nh.data_ptr())
        w_hv, w_n2, w_nh = orc.sketch_genome(s, 21, 20, 123, True, norm=norm, hv_d=512)
        assert int(nh[0]) == w_nh and int(n2[0]) == w_n2 and np.array_equal(hv[0].cpu().numpy(), w_hv), norm


def test_packed_1000x5mbp_equals_ascii_and_takes_three_eighths_of_the_memory(hg, ctx, orc):
    """BASELINE configs[1] in the packed form: 1 000 x 5 Mbp resident as blobs (1.875 GB instead of 5 GB), bit-identical
    sketches; two genomes against the oracle"""
    N, L, D = 1000, 5_000_000, 4096
    stride = (L + 1 + 15) // 16 * 16
    seq = torch.empty(N * stride + 64, dtype=torch.uint8, device="cuda")
    ctx.synth_genomes_dev(0, N, L, stride, seq.data_ptr())
    offs = np.arange(N, dtype=np.uint64) * stride
    lens = np.full(N, L + 1, np.uint64)
    bsz = hg.lib().hg_pack2_size(L + 1)
    boffs = np.arange(N, dtype=np.uint64) * bsz
    blobs = torch.empty(N * bsz + 64, dtype=torch.uint8, device="cuda")
    ctx.pack2_batch_dev(seq.data_ptr(), offs, lens, blobs.data_ptr(), boffs)
    assert blobs.numel() <= 0.376 * seq.numel()
    p = hg.default_params()
    res = []
    for form in ("ascii", "packed"):
        hv = torch.empty((N, D), dtype=torch.int16, device="cuda")
        n2 = torch.empty(N, dtype=torch.int32, device="cuda")
        nh = torch.empty(N, dtype=torch.int32, device="cuda")
        if form == "ascii":
            ctx.sketch_batch_dev(seq.data_ptr(), offs, lens, p, hv.data_ptr(), n2.data_ptr(), nh.data_ptr())
        else:
            del seq  # the ASCII copy is gone: the packed call cannot be reading it
            torch.cuda.empty_cache()
            ctx.sketch_batch_dev_packed(blobs.data_ptr(), boffs, lens, p, hv.data_ptr(), n2.data_ptr(), nh.data_ptr())
        ctx.sync()  # (reads the step's check word, then waits for the stream)
        res.append((hv, n2, nh))
    assert torch.equal(res[0][0], res[1][0]) and torch.equal(res[0][1], res[1][1]) and torch.equal(res[0][2], res[1][2])
    for g in (3, 777):
        w_hv, w_n2, w_nh = orc.sketch_genome(orc.synth_genome(g, L))
        assert int(res[1][2][g]) == w_nh and int(res[1][1][g]) == w_n2 and np.array_equal(res[1][0][g].cpu().numpy(), w_hv)


# ---- host-fed entry points: the library 2-bit packs on the host when the link is what limits ------------------------------------
# (hg_sketch_batch: batches >= 32 MB on hosts with >= 4 usable cores; one-genome calls: pageable sources or >= 3 other calls in
# flight; hook "hostfed" = "ascii" / "packed" pins the choice).  Results must not depend on the form that crossed the link.

def _host_genomes(rng, n, lo, hi):
    return [genome(rng, int(rng.integers(lo, hi))) for _ in range(n)]


@pytest.mark.parametrize("pinned", [False, True])
def test_hostfed_batch_packed_on_the_host_equals_ascii(hg, orc, pinned):
    """~300 MB in 3 sub-batches (128 MB of ASCII each when packed), ragged lengths, one empty and one shorter-than-k genome"""
    rng = np.random.default_rng(77)
    seqs = _host_genomes(rng, 70, 3_000_000, 5_500_000) + [np.zeros(0, np.uint8), genome(rng, 11)] + _host_genomes(rng, 40, 200, 90_000)
    if pinned:
        keep = [torch.from_numpy(s.copy()).pin_memory() if s.size else torch.zeros(0, dtype=torch.uint8) for s in seqs]
        seqs = [k.numpy() for k in keep]
    p = hg.default_params(hv_d=1024)
    res = {}
    for mode in ("ascii", "packed", ""):
        c = hg.Context(0)
        if mode:
            c.set_debug("hostfed", mode)
        res[mode] = c.sketch_batch(seqs, p)
        if mode == "packed":
            assert c.last_kernel("kmer").endswith("true>")
        if mode == "ascii":
            assert not c.last_kernel("kmer").endswith("true>")
        c.close()
    for mode in ("packed", ""):
        for a, b in zip(res["ascii"], res[mode]):
            assert np.array_equal(a, b), mode
    for i in (0, 69, 71, 80):
        w_hv, w_n2, w_nh = orc.sketch_genome(seqs[i], 21, 1500, 123, True, hv_d=1024)
        assert res[""][2][i] == w_nh and res[""][1][i] == w_n2 and np.array_equal(res[""][0][i], w_hv), i


@pytest.mark.parametrize("norm", [0, 1])
def test_hostfed_single_genome_calls_packed_on_the_host(hg, orc, norm):
    rng = np.random.default_rng(5 + norm)
    for n in (0, 20, 21, 1000, 300_000, 2_000_003):
        s = genome(rng, n, junk=0.01)
        if n > 100:
            s[rng.choice(n, n // 50, replace=False)] = ord("U")
        got = {}
        for mode in ("ascii", "packed", ""):  # "" = the library's choice: a pageable source >= 256 KB goes packed
            c = hg.Context(0)
            if mode:
                c.set_debug("hostfed", mode)
            got[mode] = (np.sort(c.kmer_hash_sample(s, 21, 40, 123, True, norm)), c.last_kernel("kmer"))
            if n >= 21:
                assert got[mode][1].endswith("true>") == (mode == "packed" or (mode == "" and n >= 256 * 1024)), (mode, n, got[mode][1])
            p = hg.default_params(scaled=40, hv_d=512)
            p.norm_mode = norm
            got[mode] += c.sketch_batch([s], p)
            c.close()
        want = np.sort(orc.kmer_hash_sample(s, 21, 40, 123, True, norm)) if n <= 300_000 else got["ascii"][0]
        for mode in ("ascii", "packed", ""):
            assert np.array_equal(got[mode][0], want), (mode, n)
            for a, b in zip(got[mode][2:], got["ascii"][2:]):
                assert np.array_equal(a, b), (mode, n)


def test_hostfed_concurrent_single_genome_calls(hg, orc):
    """the reference's pattern (one call per genome from a pool of threads, src/sketch_cuda.rs:79-96) from pinned memory: with
    the calls sharing the link the library packs on the calling threads; every result equals the lone ASCII call's"""
    import threading
    rng = np.random.default_rng(9)
    keep = [torch.from_numpy(genome(rng, int(rng.integers(400_000, 1_500_000)))).pin_memory() for _ in range(48)]
    seqs = [k.numpy() for k in keep]
    solo = hg.Context(0)
    solo.set_debug("hostfed", "ascii")
    want = [np.sort(solo.kmer_hash_sample(s, 21, 200)) for s in seqs]
    solo.close()
    T = 8
    ctxs = [hg.Context(0) for _ in range(T)]
    got, used_packed, bad = [None] * len(seqs), [0] * T, []

    def w(t):
        try:
            for g in range(t, len(seqs), T):
                got[g] = np.sort(ctxs[t].kmer_hash_sample(seqs[g], 21, 200))
                used_packed[t] += ctxs[t].last_kernel("kmer").endswith("true>")
        except Exception as e:  # pragma: no cover
            bad.append(repr(e))
    ths = [threading.Thread(target=w, args=(t,)) for t in range(T)]
    [x.start() for x in ths]
    [x.join() for x in ths]
    assert not bad, bad
    for g in range(len(seqs)):
        assert np.array_equal(got[g], want[g]), g
    for c in ctxs:
        c.close()


def test_hostfed_small_batches_forced_packed(hg, orc):
    """the "hostfed" = "packed" hook on batches far below the size the library would pack on its own: a few tiny genomes, an
    empty one among them, one sub-batch"""
    rng = np.random.default_rng(31)
    seqs = [genome(rng, n) for n in (1000, 0, 20, 21, 70_000, 3, 2500)]
    p = hg.default_params(scaled=20, hv_d=512)
    res = {}
    for mode in ("ascii", "packed"):
        c = hg.Context(0)
        c.set_debug("hostfed", mode)
        res[mode] = c.sketch_batch(seqs, p)
        assert c.last_kernel("kmer").endswith("true>") == (mode == "packed")
        c.close()
    for a, b in zip(res["ascii"], res["packed"]):
        assert np.array_equal(a, b)
    for i in (0, 4, 6):
        w_hv, w_n2, w_nh = orc.sketch_genome(seqs[i], 21, 20, 123, True, hv_d=512)
        assert res["packed"][2][i] == w_nh and res["packed"][1][i] == w_n2 and np.array_equal(res["packed"][0][i], w_hv)


def test_hostfed_forced_packed_with_very_short_sequences(hg, orc):
    """hg_pack2_size is 32 bytes for 1..16 bases while their padded ASCII is 16: a sub-batch dominated by such sequences has
    blobs larger than its own ASCII region of the device buffer.  Such a sub-batch must stay ASCII (its blobs would overwrite
    the next sub-batch's region or run past the buffer); one whose blobs fit goes packed as asked.  Results equal either way."""
    rng = np.random.default_rng(32)
    p = hg.default_params(ksize=5, scaled=1, hv_d=256)
    tiny = [genome(rng, int(n)) for n in rng.integers(1, 17, 3000)]
    for seqs, packs in ((tiny, False), (tiny + [genome(rng, 200_000)], True), ([genome(rng, 9)], False)):
        res = {}
        for mode in ("ascii", "packed"):
            c = hg.Context(0)
            c.set_debug("hostfed", mode)
            res[mode] = c.sketch_batch(seqs, p)
            assert c.last_kernel("kmer").endswith("true>") == (mode == "packed" and packs), (mode, len(seqs))
            c.close()
        for a, b in zip(res["ascii"], res["packed"]):
            assert np.array_equal(a, b)
        for i in (0, len(seqs) // 2, len(seqs) - 1):
            w_hv, w_n2, w_nh = orc.sketch_genome(seqs[i], 5, 1, 123, True, hv_d=256)
            assert res["packed"][2][i] == w_nh and res["packed"][1][i] == w_n2 and np.array_equal(res["packed"][0][i], w_hv)
