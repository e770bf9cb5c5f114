"""hg_pack2 (host): the 2-bit + not-a-base blob against a numpy model of the layout include/hypergen.h documents."""
import numpy as np
import pytest


@pytest.fixture(scope="module")
def hg():
    import hypergen_amd
    return hypergen_amd


def model(seq, u2t):
    seq = np.asarray(seq, np.uint8)
    n = seq.size
    up = seq & 0xDF
    ok = (up == ord("A")) | (up == ord("C")) | (up == ord("G")) | (up == ord("T"))
    if u2t:
        ok |= up == ord("U")
    code = np.where(ok, ((seq >> 1) ^ (seq >> 2)) & 3, 0).astype(np.uint8)
    cb = ((n + 3) // 4 + 15) & ~15
    mb = ((n + 7) // 8 + 15) & ~15
    c4 = np.zeros(cb * 4, np.uint8)
    c4[:n] = code
    codes = (c4[0::4] | (c4[1::4] << 2) | (c4[2::4] << 4) | (c4[3::4] << 6)).astype(np.uint8)
    m8 = np.zeros(mb * 8, np.uint8)
    m8[:n] = ~ok
    mask = np.packbits(m8, bitorder="little")
    return np.concatenate([codes, mask])


@pytest.mark.parametrize("n", [0, 1, 3, 4, 5, 15, 16, 17, 31, 32, 33, 63, 64, 65, 100, 1000, 4097, 100_003])
@pytest.mark.parametrize("u2t", [0, 1])
def test_pack2_layout(hg, n, u2t):
    rng = np.random.default_rng(n * 2 + u2t)
    seq = rng.choice(np.frombuffer(b"ACGTacgtNnUuRY-*\x00\xff ", np.uint8), n,
                     p=[.2, .2, .2, .2, .03, .03, .03, .03, .01, .01, .01, .01, .005, .005, .005, .005, .005, .005, .01])
    want = model(seq, u2t)
    assert hg.lib().hg_pack2_size(n) == want.size
    got = hg.pack2(seq, u2t)
    assert np.array_equal(got, want)
    if n:
        assert np.array_equal(hg.pack2(seq, u2t, in_place=True), want)


def test_pack2_clean_long_and_all_bytes(hg):
    rng = np.random.default_rng(5)
    seq = rng.choice(np.frombuffer(b"ACGT", np.uint8), 1_000_001)
    assert np.array_equal(hg.pack2(seq), model(seq, 0))
    every = np.arange(256, dtype=np.uint8).repeat(3)
    for u2t in (0, 1):
        assert np.array_equal(hg.pack2(every, u2t), model(every, u2t))


def decode_sparse(blob, n):
    """the hg_pack2s blob back to (codes, bitmap) of the hg_pack2 layout"""
    cb = ((n + 3) // 4 + 15) & ~15
    mb = ((n + 7) // 8 + 15) & ~15
    tab = blob[cb:].view(np.uint32)
    n_runs = int(tab[0])
    assert tab[1] == 0 and blob.size == cb + ((8 + 8 * n_runs + 15) & ~15)
    bits = np.zeros(mb * 8, np.uint8)
    prev_end = -1
    for r in range(n_runs):
        st, ln = int(tab[2 + 2 * r]), int(tab[3 + 2 * r])
        assert ln > 0 and st > prev_end and st + ln <= n  # ascending, disjoint, MAXIMAL (a gap between two runs)
        bits[st:st + ln] = 1
        prev_end = st + ln
    assert not blob[cb + 8 + 8 * n_runs:].any()  # zero padding
    return blob[:cb], np.packbits(bits, bitorder="little")


@pytest.mark.parametrize("n", [0, 1, 3, 63, 64, 65, 127, 128, 129, 1000, 4096, 4097, 100_003])
@pytest.mark.parametrize("u2t", [0, 1])
def test_pack2s_is_pack2_with_the_bitmap_as_a_run_table(hg, n, u2t):
    rng = np.random.default_rng(900 + n * 2 + u2t)
    seq = rng.choice(np.frombuffer(b"ACGTacgtNnUu", np.uint8), n, p=[.23, .23, .23, .23, .01, .01, .01, .01, .01, .01, .01, .01])
    for a, b in ((0, 3), (60, 70), (64, 128), (120, 200), (n - 5, n), (n - 64, n)):  # runs across / on word borders, at the end
        if 0 <= a < b <= n:
            seq[a:b] = ord("N")
    want = model(seq, u2t)
    cb = ((n + 3) // 4 + 15) & ~15
    blob = hg.pack2s(seq, u2t, cap=4 * n + 64)
    codes, mask = decode_sparse(blob, n)
    assert np.array_equal(codes, want[:cb]) and np.array_equal(mask, want[cb:])


def test_pack2s_extremes_and_capacity(hg):
    n = 5000
    clean = np.frombuffer(b"ACGT" * (n // 4), np.uint8)
    blob = hg.pack2s(clean)
    assert blob.size == hg.lib().hg_pack2s_size(n, 0) == ((n // 4 + 15) & ~15) + 16  # 0.25 B per base + 16
    allbad = np.full(n, ord("N"), np.uint8)
    b2 = hg.pack2s(allbad)
    assert b2[((n // 4 + 15) & ~15):].view(np.uint32)[:4].tolist() == [1, 0, 0, n]  # ONE run
    alt = np.frombuffer(b"AN" * (n // 2), np.uint8)  # n / 2 runs: 4 n bytes of table -- does not fit the bitmap's size
    assert hg.pack2s(alt) is None
    got = hg.pack2s(alt, cap=8 * n)
    codes, mask = decode_sparse(got, n)
    assert np.array_equal(np.concatenate([codes, mask]), model(alt, 0))
