// hg_dist_tile_order.h -- which workgroup slot runs which tile: the host-built slot -> tile table of a GEMM launch
// (private to hg_dist_kernels.hip; pure host code).
#pragma once
#include <algorithm>

#include "hg_dist_gemm.h"

namespace {

// The order itself (pure host code, no device involved; may throw std::bad_alloc): slot b -> tm | tn << 16, ~0u = no tile.
static std::vector<uint32_t> build_tile_order(uint32_t tiles_m, uint32_t tiles_n, uint32_t bm, uint32_t bn, bool diag, bool symmetric,
                                              uint64_t ref_off, uint64_t qry_off) {
  std::vector<uint32_t> host;
  std::vector<uint32_t> dg, walk, group;  // group[i]: the half super-tile (4 x 8 tiles) walk[i] belongs to
  auto has_work = [&](uint32_t tm, uint32_t tn) {
    return !(symmetric && (uint64_t)tm * bm + ref_off >= (uint64_t)tn * bn + qry_off + bn);
  };
  auto on_diag = [&](uint32_t tm, uint32_t tn) { return diag && (tn == tm * bm / bn || tn == (tm * bm + bm - 1) / bn); };
  if (diag)
    for (uint32_t second = 0; second < 2; ++second)  // the rows' first diagonal tiles, the dense ones, go round the XCDs first
      for (uint32_t tm = 0; tm < tiles_m; ++tm) {
        const uint32_t tn0 = tm * bm / bn, tn1 = (tm * bm + bm - 1) / bn, tn = second ? tn1 : tn0;
        if ((second && tn1 == tn0) || tn >= tiles_n || !has_work(tm, tn)) continue;
        dg.push_back(tm | tn << 16);
      }
  const uint32_t sup_m = (tiles_m + ST - 1) / ST, sup_n = (tiles_n + ST - 1) / ST;
  for (uint32_t sup = 0; sup < sup_m * sup_n; ++sup)
    for (uint32_t within = 0; within < ST * ST; ++within) {
      const uint32_t tm = (sup / sup_n) * ST + within / ST, tn = (sup % sup_n) * ST + within % ST;
      if (tm >= tiles_m || tn >= tiles_n || on_diag(tm, tn) || !has_work(tm, tn)) continue;
      walk.push_back(tm | tn << 16);
      group.push_back(2 * sup + within / (ST * ST / 2));
    }
  // the queue of XCD x: its share of the diagonal tiles, then one contiguous run of the walk
  std::vector<uint32_t> queue[8];
  for (size_t i = 0; i < dg.size(); ++i) queue[i % 8].push_back(dg[i]);
  const size_t total = dg.size() + walk.size(), q = total / 8, r = total % 8;
  // The 32 workgroups resident on an XCD are a window of its queue: it should lie on ONE half super-tile (4 A blocks,
  // 8 B blocks) as long as possible, so the half super-tiles that a run holds only in part -- at most its first and
  // its last -- go to the END of the queue and the whole ones keep their phase (the Hamming search at 50 000 x 10 000
  // x 16384 moved 8.6 GB through the L2s with the runs cut wherever the count said, 7.0 GB before the table existed).
  size_t w = 0;
  for (size_t x = 0; x < 8; ++x) {
    const size_t mine = q + (x < r ? 1 : 0), w0 = w;
    size_t w1 = w0;
    for (size_t have = queue[x].size(); have < mine && w1 < walk.size(); ++have) ++w1;
    std::vector<uint32_t> part;
    for (size_t i = w0; i < w1; ++i) {
      const bool head = w0 > 0 && group[i] == group[w0 - 1], tail = w1 < walk.size() && group[i] == group[w1];
      (head || tail ? part : queue[x]).push_back(walk[i]);
    }
    queue[x].insert(queue[x].end(), part.begin(), part.end());
    w = w1;
  }
  for (size_t x = 0; w < walk.size(); x = (x + 1) % 8) queue[x].push_back(walk[w++]);  // (tiny grids only: a share smaller than its diagonal tiles)
  size_t rows = 0;
  for (auto &qu : queue) rows = std::max(rows, qu.size());
  host.assign(std::max<size_t>(rows, 1) * 8, ~0u);
  for (size_t x = 0; x < 8; ++x)
    for (size_t j = 0; j < queue[x].size(); ++j) host[8 * j + x] = queue[x][j];
  return host;
}
static uint32_t dist_tile_table(hg_ctx *c, GemmArgs &g, uint32_t bm, uint32_t bn, bool diag) {
  const uint32_t legacy_diag = diag ? 2 * ((g.tiles_m + 7) / 8 * 8) : 0u;
  auto legacy = [&]() {
    g.tile_tab = nullptr, g.diag_first = legacy_diag;
    return legacy_diag + dist_grid(g.tiles_m, g.tiles_n);
  };
  if (c->dbg_dist_order == "legacy" || g.tiles_m > 0xFFFFu || g.tiles_n > 0xFFFEu || (uint64_t)g.tiles_m * g.tiles_n > (1u << 24)) return legacy();
  const uint32_t flags = (diag ? 1u : 0u) | (g.symmetric ? 2u : 0u);
  hg_ctx::TileTab *hit = nullptr, *lru = &c->tile_tabs[0];
  for (auto &t : c->tile_tabs) {
    if (t.n_slots && t.tiles_m == g.tiles_m && t.tiles_n == g.tiles_n && t.bm == bm && t.bn == bn && t.flags == flags &&
        (!g.symmetric || t.ref_off - t.qry_off == (uint64_t)g.ref_off - (uint64_t)g.qry_off))  // (the triangle test sees only the difference)
      hit = &t;
    if (t.used < lru->used) lru = &t;
  }
  if (!hit) {
    hg_ctx::TileTab &t = *lru;
    // an evicted table: its device copy is rewritten in stream order behind the launches that read it; its host copy
    // once the old upload has passed
    if (!t.uploaded && hipEventCreateWithFlags(&t.uploaded, hipEventDisableTiming) != hipSuccess) {
      (void)hipGetLastError();
      t.uploaded = nullptr;
      return legacy();
    }
    if (t.used) (void)hipEventSynchronize(t.uploaded);
    t.n_slots = 0;
    try {
      t.host = build_tile_order(g.tiles_m, g.tiles_n, bm, bn, diag, g.symmetric != 0, g.ref_off, g.qry_off);
    } catch (const std::bad_alloc &) {
      t.n_slots = 0;
      return legacy();
    }
    if (hg_ensure(c, t.dev, t.host.size() * sizeof(uint32_t)) != HG_OK) {
      t.n_slots = 0;
      return legacy();
    }
    // (ordered on the ctx's stream like every other workspace write: a launch that still reads the evicted table is ahead of it)
    if (hipMemcpyAsync(t.dev.p, t.host.data(), t.host.size() * sizeof(uint32_t), hipMemcpyHostToDevice, c->stream) != hipSuccess) {
      (void)hipGetLastError();
      t.n_slots = 0;
      return legacy();
    }
    (void)hipEventRecord(t.uploaded, c->stream);
    t.tiles_m = g.tiles_m, t.tiles_n = g.tiles_n, t.bm = bm, t.bn = bn, t.flags = flags, t.ref_off = g.ref_off, t.qry_off = g.qry_off;
    t.n_slots = (uint32_t)t.host.size();
    hit = &t;
  }
  hit->used = ++c->tile_tab_clock;
  g.tile_tab = static_cast<const uint32_t *>(hit->dev.p), g.diag_first = 0;
  return hit->n_slots;
}

}  // namespace
