"""The batch plan of the sketch step (hyper-gen_amd/csrc/hg_sketch_plan.hip) as host arithmetic -- no GPU: hit regions sized for
twice the expected sample + 1 024 (src/sketch.rs:73: one k-mer in `scaled`), work items of one genome each, and since round 6
the work items of consecutive SMALL genomes grouped into one workgroup of the k-mer launch."""
import numpy as np
import pytest


@pytest.fixture(scope="module")
def hg():
    import hypergen_amd
    return hypergen_amd


def offsets_for(lens):
    return np.concatenate([[0], np.cumsum((np.asarray(lens, np.uint64) + 15) // 16 * 16)[:-1]]).astype(np.uint64)


def tiles_of_items(lens, k, item, tile):
    out = []
    for L in lens:
        starts = max(0, int(L) - k + 1)
        n_items = (starts + item - 1) // item
        for it in range(n_items):
            out.append((min(item, starts - it * item) + tile - 1) // tile)
    return out


def check_groups(hg, lens, k=21, scaled=1500):
    d, gf = hg.sketch_plan_describe(offsets_for(lens), lens, k, scaled)
    item = 27432 if k <= 21 else 27324
    tile = item // 9
    tiles = tiles_of_items(lens, k, item, tile)
    assert d["items"] == len(tiles) and d["item_tiles"] == 9
    assert gf[0] == 0 and gf[-1] == d["items"] and (np.diff(gf.astype(np.int64)) > 0).all()  # every item in exactly one workgroup
    for a, b in zip(gf[:-1], gf[1:]):
        grp = tiles[a:b]
        if len(grp) > 1:
            assert sum(grp) <= 27 and max(grp) < 9, (a, b, grp)  # small items only, three full items' worth of tiles at most
    return d, gf, tiles


def test_large_genomes_keep_one_workgroup_per_item(hg):
    lens = np.full(1000, 5_000_001, np.uint64)
    d, gf, tiles = check_groups(hg, lens)
    assert d["items"] == 1000 * 183 and d["workgroups"] == d["items"]  # 182 full items + a 3-tile rest per genome, none grouped
    assert d["max_expect"] == (5_000_001 - 20) // 1500 and d["max_cap"] == 2 * d["max_expect"] + 1024
    assert d["hit_slots"] == 1000 * d["max_cap"]


def test_small_genomes_share_workgroups(hg):
    d, gf, tiles = check_groups(hg, np.full(400_000, 2001, np.uint64))
    assert d["items"] == 400_000 and d["workgroups"] == (400_000 + 26) // 27  # one tile each: 27 genomes per workgroup
    d, gf, tiles = check_groups(hg, np.full(200_000, 10_001, np.uint64))
    assert set(tiles) == {4} and d["workgroups"] == (200_000 + 5) // 6  # four tiles each: six genomes per workgroup
    assert d["max_cap"] == 2 * ((10_001 - 20) // 1500) + 1024


def test_mixed_batch_groups_only_the_small_items(hg):
    rng = np.random.default_rng(5)
    lens = rng.choice([0, 5, 20, 21, 300, 3048 + 20, 3049 + 20, 27432 + 20, 27433 + 20, 60_000, 400_000, 2_000_000], 3000).astype(np.uint64)
    d, gf, tiles = check_groups(hg, lens)
    assert d["workgroups"] < d["items"]
    full = [i for i, t in enumerate(tiles) if t == 9]
    starts = set(int(x) for x in gf[:-1])
    assert all(i in starts and (i + 1) in starts | {d["items"]} for i in full)  # an item that fills its nine tiles is alone
    for k in (25, 32):
        check_groups(hg, lens, k=k, scaled=200)


def test_long_k_has_no_groups_and_genomes_shorter_than_k_have_no_items(hg):
    lens = np.array([10, 32, 33, 34, 12288 + 32, 12289 + 32, 100_000], np.uint64)
    d, gf = hg.sketch_plan_describe(offsets_for(lens), lens, 33, 1500)
    assert gf is None and d["item_tiles"] == 0 and d["workgroups"] == d["items"] == 0 + 0 + 1 + 1 + 1 + 2 + 9
    d, gf = hg.sketch_plan_describe(offsets_for(lens), lens, 21, 1)
    assert d["max_cap"] == 131072  # scaled = 1: at most one slot per k-mer (99 981), rounded to a power of two beyond the LDS sort


def test_plan_rejects_misaligned_offsets(hg):
    with pytest.raises(hg.HgError) as e:
        hg.sketch_plan_describe(np.array([0, 1002], np.uint64), np.array([1000, 1000], np.uint64))
    assert e.value.status == hg.ERR_INVALID
