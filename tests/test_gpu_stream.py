"""hg_sketch_stream_* (continuous host-fed sketching: uploader + compute thread per device) against hg_sketch_batch
and the oracle: same bits whatever the chunking, the pushing thread, the memory kind or the number of engines."""
import threading

import numpy as np
import pytest

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def hg():
    import hypergen_amd
    return hypergen_amd


def _drain(st, n):
    out = {}
    while True:
        r = st.pop()
        if r is None:
            break
        assert r[0] not in out
        out[r[0]] = r[1:]
    assert len(out) == n
    return out


def test_stream_equals_batch_mixed_sizes(hg, orc):
    # empty, shorter than k, exactly k, tiny (staged together), mid-size, 5 Mbp, and one genome larger than a chunk
    lens = [0, 10, 21, 22, 2_000, 2_000, 30_000, 255_000, 300_000, 1_200_000, 5_000_000, 70_000_000, 4_000, 900]
    genomes = [orc.synth_genome(g, L)[1:] if L else np.zeros(0, np.uint8) for g, L in enumerate(lens)]
    genomes[6] = genomes[6].copy()
    genomes[6][1000:1040] = ord("N")
    p = hg.default_params(scaled=200)
    with hg.Context(0) as ctx:
        hv, n2, nh = ctx.sketch_batch(genomes, p)
    for devs in ((0,), (0, 0)):
        with hg.SketchStream(devs, p) as st:
            def pusher(idx):
                for i in idx:
                    st.push(genomes[i], i)
            th = [threading.Thread(target=pusher, args=(range(t, len(genomes), 3),)) for t in range(3)]
            for t in th:
                t.start()
            for t in th:
                t.join()
            st.finish()
            out = _drain(st, len(genomes))
        for i in range(len(genomes)):
            assert out[i][2] == nh[i] and out[i][1] == n2[i] and np.array_equal(out[i][0], hv[i]), (devs, i)
    for i in (4, 7):
        w_hv, w_n2, w_nh = orc.sketch_genome(genomes[i], scaled=200)
        assert nh[i] == w_nh and n2[i] == w_n2 and np.array_equal(hv[i], w_hv)


def test_stream_many_small_genomes_and_interleaved_pops(hg, orc):
    """more genomes than one chunk may hold (4096), pops interleaved with pushes, page-locked and plain memory"""
    import torch
    n = 6000
    base = orc.synth_genome(7, 400_000)[1:]
    rng = np.random.default_rng(3)
    starts = rng.integers(0, base.size - 3000, n)
    sizes = rng.integers(500, 3000, n)
    genomes = [base[s:s + m] for s, m in zip(starts, sizes)]
    pinned = torch.from_numpy(base.copy()).pin_memory().numpy()
    p = hg.default_params(scaled=20)
    with hg.Context(0) as ctx:
        hv, n2, nh = ctx.sketch_batch(genomes, p)
    with hg.SketchStream((0,), p) as st:
        got = {}
        for i in range(n):
            src = pinned if i % 2 else base
            st.push(src[starts[i]:starts[i] + sizes[i]], i)
            if i % 1000 == 999:  # results outstanding: pop some while pushing
                for _ in range(500):
                    r = st.pop()
                    got[r[0]] = r[1:]
        st.finish()
        got.update(_drain(st, n - len(got)))
    assert len(got) == n
    for i in range(n):
        assert got[i][2] == nh[i] and got[i][1] == n2[i] and np.array_equal(got[i][0], hv[i]), i


def test_stream_params_and_errors(hg, orc):
    g = orc.synth_genome(3, 200_000)[1:]
    p = hg.default_params(ksize=16, scaled=50, hv_d=1024, canonical=0)
    with hg.Context(0) as ctx:
        hv, n2, nh = ctx.sketch_batch([g], p)
    with hg.SketchStream((0,), p) as st:
        st.push(g, 77)
        st.finish()
        tag, shv, sn2, snh = st.pop()
        assert tag == 77 and snh == nh[0] and sn2 == n2[0] and np.array_equal(shv, hv[0])
        assert st.pop() is None and st.pop() is None
        with pytest.raises(hg.HgError):
            st.push(g, 1)  # after finish
    bad = hg.default_params(hv_d=0)
    with pytest.raises(hg.HgError):
        hg.SketchStream((0,), bad)
    with pytest.raises(hg.HgError):
        hg.SketchStream((99,), p)


def _dirty(orc, g, L, rng):
    seq = orc.synth_genome(g, L)[1:].copy() if L else np.zeros(0, np.uint8)
    if L > 100:
        seq[rng.choice(L, max(1, L // 5000), replace=False)] = ord("N")
        a = int(rng.integers(0, L - 60))
        seq[a:a + 50] = np.char.lower(seq[a:a + 50].view("S1")).view(np.uint8)
        b = int(rng.integers(0, L - 60))
        seq[b:b + 40] = rng.choice(np.frombuffer(b"UuRYKM-*", np.uint8), 40)
    return seq


@pytest.mark.parametrize("norm", [0, 1])
def test_unpack2_dev_is_the_normalised_ascii(hg, orc, norm):
    """pack on the host, expand on the device: 'A','C','G','T' where the kernels see a base (u/U -> T under U2T), 'N' elsewhere"""
    import torch
    rng = np.random.default_rng(11 + norm)
    with hg.Context(0) as ctx:
        for L in (1, 15, 16, 17, 4095, 4096, 16_385, 1_000_003):
            seq = _dirty(orc, 5, L, rng) if L > 100 else rng.choice(np.frombuffer(b"ACGTNacgtU", np.uint8), L)
            blob = torch.from_numpy(hg.pack2(seq, norm)).cuda()
            out = torch.full(((L + 15) // 16 * 16,), 0x7F, dtype=torch.uint8, device="cuda")
            ctx.unpack2_dev(blob.data_ptr(), L, out.data_ptr())
            ctx.sync()
            up = seq & 0xDF
            ok = (up == 65) | (up == 67) | (up == 71) | (up == 84) | ((up == 85) if norm else np.zeros(L, bool))
            want = np.where(ok, np.where(up == 85, 84, up), ord("N")).astype(np.uint8)
            assert np.array_equal(out.cpu().numpy()[:L], want), L


@pytest.mark.parametrize("norm", [0, 1])
def test_stream_packed_equals_ascii(hg, orc, norm):
    rng = np.random.default_rng(21 + norm)
    lens = [0, 5, 21, 40, 3_000, 100_000, 777_777, 5_000_000, 70_000_000, 2_222]
    genomes = [_dirty(orc, 30 + i, L, rng) for i, L in enumerate(lens)]
    p = hg.default_params(scaled=300, norm_mode=norm)
    with hg.Context(0) as ctx:
        hv, n2, nh = ctx.sketch_batch(genomes, p)
    with hg.SketchStream((0,), p) as st:
        for i, g in enumerate(genomes):
            if i % 3 == 2:
                st.push(g, i)  # mixed chunks: ASCII and packed genomes side by side
            else:
                st.push_packed(hg.pack2(g, norm), g.size, i)
        st.finish()
        out = _drain(st, len(genomes))
    for i in range(len(genomes)):
        assert out[i][2] == nh[i] and out[i][1] == n2[i] and np.array_equal(out[i][0], hv[i]), i
    w_hv, w_n2, w_nh = orc.sketch_genome(genomes[5], scaled=300, norm=norm)
    assert nh[5] == w_nh and n2[5] == w_n2 and np.array_equal(hv[5], w_hv)


@pytest.mark.parametrize("norm", [0, 1])
def test_stream_sparse_packed_equals_ascii(hg, orc, norm):
    """hg_pack2s over the link (codes + run table, the bitmap rebuilt on the device): chunks of sparse blobs only, and
    chunks that mix all three forms; a genome littered with non-bases, one that is ONE run, runs on slice borders"""
    rng = np.random.default_rng(61 + norm)
    lens = [0, 5, 21, 40, 3_000, 32_768, 100_000, 777_777, 5_000_000, 20_000_000, 2_222, 65_536, 131_073]
    genomes = [_dirty(orc, 50 + i, L, rng) for i, L in enumerate(lens)]
    genomes[5][:] = ord("N")                         # nothing but one run: exactly one bitmap slice of ones
    genomes[6][32_760:32_776] = ord("n")             # a run across the border of two slices
    genomes[11][np.arange(0, 65_536, 7)] = ord("-")  # ~9 400 runs: still fits the bitmap's size as a table? (no: falls back)
    p = hg.default_params(scaled=100, norm_mode=norm)
    with hg.Context(0) as ctx:
        hv, n2, nh = ctx.sketch_batch(genomes, p)
    for mixed in (False, True):
        with hg.SketchStream((0,), p) as st:
            n_sparse = 0
            for i, g in enumerate(genomes):
                blob = hg.pack2s(g, norm)
                if mixed and i % 3 == 1:
                    st.push(g, i)
                elif blob is None or (mixed and i % 3 == 2):
                    st.push_packed(hg.pack2(g, norm), g.size, i)
                else:
                    st.push_packed_sparse(blob, g.size, i)
                    n_sparse += 1
            st.finish()
            out = _drain(st, len(genomes))
        assert n_sparse >= (4 if mixed else 10)
        for i in range(len(genomes)):
            assert out[i][2] == nh[i] and out[i][1] == n2[i] and np.array_equal(out[i][0], hv[i]), (mixed, i)
    w_hv, w_n2, w_nh = orc.sketch_genome(genomes[6], scaled=100, norm=norm)
    assert nh[6] == w_nh and n2[6] == w_n2 and np.array_equal(hv[6], w_hv)


def test_sparse_blobs_are_validated_before_they_reach_the_device(hg, orc):
    """hg_pack2s blobs carry their own run count and table: a short buffer, a count that would read past it, and a table
    that is not ascending / disjoint / inside the sequence are refused on the host (the device's binary search over the
    table assumes all of that); the stream stays usable"""
    import ctypes as C
    rng = np.random.default_rng(71)
    g = _dirty(orc, 3, 200_000, rng)
    blob = hg.pack2s(g)
    assert blob is not None
    tab = (((g.size + 3) // 4) + 15) & ~15
    n_runs = int(blob[tab: tab + 4].view("<u4")[0])
    assert n_runs >= 3
    p = hg.default_params(scaled=100)
    with hg.Context(0) as ctx:
        hv, n2, nh = ctx.sketch_batch([g], p)
    with hg.SketchStream((0,), p) as st:
        with pytest.raises(hg.HgError):
            st.push_packed_sparse(blob[: tab + 16], g.size, 0)  # the wrapper's size check
        bad = []
        b1 = blob.copy()  # runs swapped: not ascending
        b1[tab + 8: tab + 16], b1[tab + 16: tab + 24] = blob[tab + 16: tab + 24].copy(), blob[tab + 8: tab + 16].copy()
        bad.append(b1)
        b2 = blob.copy()  # a run that reaches past the sequence
        b2[tab + 8 + 8 * (n_runs - 1): tab + 16 + 8 * (n_runs - 1)].view("<u4")[:] = [g.size - 2, 5]
        bad.append(b2)
        b3 = blob.copy()  # an empty run
        b3[tab + 12: tab + 16].view("<u4")[:] = 0
        bad.append(b3)
        for i, b in enumerate(bad):
            with pytest.raises(hg.HgError) as ei:
                st.push_packed_sparse(b, g.size, 10 + i)
            assert ei.value.status == hg.ERR_INVALID
        # the sized C entry point: the buffer's true size is part of the call
        L = hg.lib()
        assert L.hg_sketch_stream_push_packed_sparse(st._h, hg._ptr(blob), tab + 4, g.size, C.c_uint64(77)) == hg.ERR_INVALID
        assert L.hg_sketch_stream_push_packed_sparse(st._h, hg._ptr(blob), blob.size - 1, g.size, C.c_uint64(77)) == hg.ERR_INVALID
        st.push_packed_sparse(blob, g.size, 5)
        st.finish()
        out = _drain(st, 1)
        assert out[5][1] == n2[0] and out[5][2] == nh[0] and np.array_equal(out[5][0], hv[0])


def test_read_fastx_pinned_pack_flag(hg, orc, tmp_path):
    seq = _dirty(orc, 9, 200_000, np.random.default_rng(2))
    f = tmp_path / "g.fna"
    body = seq.tobytes()
    f.write_bytes(b">g\n" + b"\n".join(body[j:j + 70] for j in range(0, len(body), 70)) + b"\n")
    merged = hg.read_merge_seq(str(f))
    with hg.PinnedReader() as rd:
        for mode, norm in ((hg.READ_MERGE | 16, 0), (hg.READ_NEEDLETAIL | 16 | 32, 1)):
            blob = rd.read(str(f), mode=mode)
            n = merged.size
            assert blob.size == n  # the view is n_bps long; the blob occupies its first hg_pack2_size(n) bytes
            size = hg.lib().hg_pack2_size(n)
            assert np.array_equal(blob[:size], hg.pack2(merged, norm))


def test_stream_concurrent_push_pop_and_early_close(hg, orc):
    """four pushers and two poppers at once; then a stream that is closed with results outstanding must not hang"""
    base = orc.synth_genome(12, 600_000)[1:]
    rng = np.random.default_rng(8)
    n = 3000
    starts = rng.integers(0, base.size - 20_000, n)
    sizes = rng.integers(30, 20_000, n)
    p = hg.default_params(scaled=50)
    genomes = [base[s:s + m] for s, m in zip(starts, sizes)]
    with hg.Context(0) as ctx:
        hv, n2, nh = ctx.sketch_batch(genomes, p)
    got, lock = {}, threading.Lock()
    with hg.SketchStream((0, 0), p) as st:
        def pusher(t):
            for i in range(t, n, 4):
                if i % 5 == 0:
                    st.push_packed(hg.pack2(genomes[i]), genomes[i].size, i)
                else:
                    st.push(genomes[i], i)

        def popper():
            while True:
                with lock:
                    if len(got) >= n:
                        return
                r = st.pop()
                if r is None:
                    return
                with lock:
                    got[r[0]] = r[1:]
        pu = [threading.Thread(target=pusher, args=(t,)) for t in range(4)]
        po = [threading.Thread(target=popper) for _ in range(2)]
        for t in pu + po:
            t.start()
        for t in pu:
            t.join()
        st.finish()
        for t in po:
            t.join()
    assert len(got) == n
    for i in range(n):
        assert got[i][2] == nh[i] and got[i][1] == n2[i] and np.array_equal(got[i][0], hv[i]), i
    st = hg.SketchStream((0,), p)
    for i in range(200):
        st.push(genomes[i], i)
    st.pop()
    st.close()  # 199 results dropped


def test_single_thread_push_everything_then_pop_and_reused_tags(hg, orc):
    """One thread pushes more genomes than the stream keeps outstanding (4096) before it pops anything -- the blocking C
    push would wait for ever there; hg_sketch_stream_try_push reports "would block" and the wrapper drains.  Tags are the
    caller's business: the same tag for every genome must neither lose results nor drop an array that is still read."""
    n = 4500
    base = orc.synth_genome(3, 40_000)[1:]
    genomes = [base[i: i + 3_000 + (i % 7)].copy() for i in range(n)]
    p = hg.default_params(scaled=50)
    with hg.Context(0) as ctx:
        hv, n2, nh = ctx.sketch_batch(genomes, p)
    lib = hg.lib()
    with hg.SketchStream((0,), p) as st:
        assert lib.hg_sketch_stream_max_pending(st._h) == 4096
        for i in range(n):
            st.push(genomes[i], i)  # would dead-lock at i = 4096 with the blocking entry point
        st.finish()
        out = _drain(st, n)
    for i in range(0, n, 97):
        assert out[i][2] == nh[i] and out[i][1] == n2[i] and np.array_equal(out[i][0], hv[i]), i
    # the raw entry point: fill the stream, then one more push must say HG_ERR_CAPACITY at once
    with hg.SketchStream((0,), p) as st:
        big = [orc.synth_genome(g, 6_000_000)[1:] for g in range(2)]
        pending = 0
        import ctypes as C
        st_code = hg.OK
        tiny = np.zeros(4, np.uint8)
        while st_code == hg.OK and pending < 5000:
            a = big[pending % 2] if pending < 2 else tiny
            st_code = lib.hg_sketch_stream_try_push(st._h, C.c_void_p(a.ctypes.data), a.size, pending, 0)
            pending += st_code == hg.OK
        assert st_code == hg.ERR_CAPACITY and pending == 4096
        got = C.c_int()
        for _ in range(pending):
            assert lib.hg_sketch_stream_pop(st._h, None, None, None, None, C.byref(got)) == hg.OK and got.value == 1
    # one tag for everything
    with hg.SketchStream((0,), p) as st:
        for i in range(300):
            st.push(genomes[i].copy(), 7)  # temporaries: only the wrapper keeps them alive
        st.finish()
        res = []
        while True:
            r = st.pop()
            if r is None:
                break
            res.append(r)
        assert len(res) == 300 and all(r[0] == 7 for r in res)
        assert sorted(int(r[3]) for r in res) == sorted(int(x) for x in nh[:300])
