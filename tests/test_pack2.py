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
