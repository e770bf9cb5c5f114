"""The slot -> tile table of the dist / Hamming GEMM launches (hg_dist_tile_order, host code: no GPU needed).  A tile that is
missing from the table is a block of the ANI matrix that silently reports no hits, so the invariants are checked over many
shapes: every tile that has work exactly once, no other tile, the XCDs (slot b -> XCD b % 8) balanced to one tile, diagonal
tiles in front, whole half super-tiles contiguous in an XCD's queue."""
import numpy as np
import pytest

import hypergen_amd as hg

NONE = 0xFFFFFFFF


def expected_tiles(tm_n, tn_n, bm, bn, symmetric, ref_off, qry_off):
    want = set()
    for tm in range(tm_n):
        for tn in range(tn_n):
            if symmetric and tm * bm + ref_off >= tn * bn + qry_off + bn:
                continue  # entirely on / below the diagonal: row i and column j only count for i + ref_off < j + qry_off
            want.add((tm, tn))
    return want


def on_diag(tm, tn, bm, bn):
    return tn == tm * bm // bn or tn == (tm * bm + bm - 1) // bn


SHAPES = [(40, 32, 256, 320), (40, 40, 256, 256), (28, 22, 256, 320), (196, 32, 256, 320), (1, 1, 256, 320), (1, 9, 256, 256),
          (9, 1, 256, 256), (3, 5, 128, 128), (79, 63, 256, 320), (8, 8, 256, 256), (17, 33, 256, 192), (53, 53, 256, 192),
          (2, 40, 256, 320), (100, 7, 128, 128)]


@pytest.mark.parametrize("shape", SHAPES)
@pytest.mark.parametrize("diag", [False, True])
@pytest.mark.parametrize("sym", [(False, 0, 0), (True, 0, 0), (True, 2560, 0), (True, 0, 5120), (True, 777, 1234)])
def test_table_covers_every_tile_once_and_balances_the_xcds(shape, diag, sym):
    tm_n, tn_n, bm, bn = shape
    symmetric, ref_off, qry_off = sym
    tab = hg.dist_tile_order(tm_n, tn_n, bm, bn, diag, symmetric, ref_off, qry_off)
    assert tab.size % 8 == 0 and tab.size >= 8
    real = tab[tab != NONE]
    got = [(int(t) & 0xFFFF, int(t) >> 16) for t in real]
    want = expected_tiles(tm_n, tn_n, bm, bn, symmetric, ref_off, qry_off)
    assert len(got) == len(set(got)), "a tile appears twice"
    assert set(got) == want
    per_xcd = [(tab[x::8] != NONE).sum() for x in range(8)]
    assert max(per_xcd) - min(per_xcd) <= 1, per_xcd
    assert tab.size == 8 * max(max(per_xcd), 1)  # no slot rows beyond the longest queue
    for x in range(8):  # a queue has no holes: its empty slots are at its end
        q = tab[x::8]
        k = (q != NONE).sum()
        assert (q[:k] != NONE).all() and (q[k:] == NONE).all()
    if diag and len(want) >= 64:
        # in every XCD's queue the tiles on the diagonal come before all others
        for x in range(8):
            q = [(int(t) & 0xFFFF, int(t) >> 16) for t in tab[x::8] if t != NONE]
            flags = [on_diag(tm, tn, bm, bn) for tm, tn in q]
            assert flags == sorted(flags, reverse=True), (x, flags[:20])


def test_whole_half_super_tiles_stay_together():
    """50 000 x 10 000 Hamming search: 196 x 32 tiles; every XCD's queue consists of whole 4 x 8 blocks except at its end"""
    tab = hg.dist_tile_order(196, 32, 256, 320)
    for x in range(8):
        q = [int(t) for t in tab[x::8] if t != NONE]
        blocks = [((t & 0xFFFF) // 4, (t >> 16) // 8) for t in q]
        # count the positions where the block changes: a queue of n tiles in whole blocks of 32 has n / 32 - 1 changes
        changes = sum(1 for a, b in zip(blocks, blocks[1:]) if a != b)
        assert changes <= len(q) // 32 + 2, (x, changes, len(q))
        whole = 0
        for i in range(0, len(q) - 31, 32):
            whole += len(set(blocks[i:i + 32])) == 1
        assert whole >= len(q) // 32 - 2, (x, whole)


def test_bench_shape_gives_every_cu_five_tiles():
    """10 000 x 10 000 at 256 x 320: 1 280 tiles, 160 per XCD = 5 per CU, no empty slot anywhere -- with or without the
    diagonal in front (the blockIdx mapping it replaces had 80 slots that returned at once: 155..165 tiles per XCD)"""
    for diag in (False, True):
        tab = hg.dist_tile_order(40, 32, 256, 320, diag)
        assert tab.size == 1280 and (tab != NONE).all()
    tab = hg.dist_tile_order(40, 32, 256, 320, True)
    head = [(int(t) & 0xFFFF, int(t) >> 16) for t in tab[:40]]
    assert all(on_diag(tm, tn, 256, 320) for tm, tn in head) and sorted(tm for tm, _ in head) == list(range(40))


def test_bad_arguments():
    with pytest.raises(hg.HgError):
        hg.dist_tile_order(70000, 3)
    with pytest.raises(hg.HgError):
        hg.dist_tile_order(5, 5, 0, 320)
    n = hg.C.c_size_t(0)
    small = np.zeros(4, np.uint32)
    assert hg.lib().hg_dist_tile_order(40, 32, 256, 320, 0, 0, 0, 0, hg.C.c_void_p(small.ctypes.data), 4, hg.C.byref(n)) == hg.ERR_CAPACITY
    assert n.value == 1280
