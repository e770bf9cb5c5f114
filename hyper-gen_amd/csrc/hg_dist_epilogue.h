// hg_dist_epilogue.h -- what happens to a finished accumulator tile of dist_mfma_kernel: the row / column words staged at
// kernel entry, the threshold pre-filter over the accumulators, the candidate lists in LDS, the exact reference arithmetic
// (src/dist.rs:153-160) and the hit list.  Private to hg_dist_kernels.hip.
#pragma once
#include "hg_dist_gemm.h"

namespace {

// What a thread fetches for the tile's rows / columns at kernel entry (one row or column per thread and pass)
template <bool BIG, int NT>
struct DistTileWords {
  static constexpr int WORD_PASSES = (TileCfg<BIG, NT>::BM + TileCfg<BIG, NT>::BN + TileCfg<BIG, NT>::THREADS - 1) / TileCfg<BIG, NT>::THREADS;
  int32_t w_nv[WORD_PASSES], w_info[WORD_PASSES], w_slot[WORD_PASSES];
  uint32_t w_first[WORD_PASSES];
};

// The tile's row / column words for the epilogue are staged NOW: their global loads are issued in front of the
// first operand tile's, travel with it, and the barrier below publishes what is computed from them (fetched after
// the K loop they cost a dependent-load latency per tile with nothing to hide it behind).  Per row / column: the
// norm, the i8 path's info / outlier words, and the phase-0 threshold (see the epilogue), so that the accumulator
// sweep reads ONE float per row and column.
// dist_load_tile_words issues the global loads (call it in front of the main loop), dist_stage_tile_words computes the
// thresholds and stores everything to LDS (call it from the main loop's publish_tile_words hook).
template <bool BIG, int NT, bool I8, bool HAM, bool CEN>
__device__ __forceinline__ void dist_load_tile_words(const GemmArgs &g, uint32_t row0, uint32_t col0, DistTileWords<BIG, NT> &w_) {
  using TC = TileCfg<BIG, NT>;
  constexpr int BM = TC::BM, BN = TC::BN, THREADS = TC::THREADS, WORD_PASSES = DistTileWords<BIG, NT>::WORD_PASSES;
  constexpr bool CENT = (I8 && !HAM) || CEN;
  const uint32_t tid = threadIdx.x;
  auto &w_nv = w_.w_nv, &w_info = w_.w_info, &w_slot = w_.w_slot;
  auto &w_first = w_.w_first;
#pragma unroll
  for (int p = 0; p < WORD_PASSES; ++p) {
    const uint32_t t = tid + (uint32_t)p * THREADS;
    const bool is_r = t < (uint32_t)BM;
    const uint32_t idx = is_r ? row0 + t : col0 + (t - BM);
    const bool in = t < (uint32_t)(BM + BN) && idx < (is_r ? g.R : g.Q);
    w_nv[p] = (in && !HAM) ? (is_r ? g.nr[idx] : g.nq[idx]) : 0;
    w_info[p] = w_slot[p] = 0, w_first[p] = 0u;
    if (CENT) w_info[p] = in ? (is_r ? g.info_r[idx] : g.info_q[idx]) : 0;
    if (I8 && !HAM) {
      w_slot[p] = in ? (is_r ? g.slot_r[idx] : g.slot_q[idx]) : 0;
      w_first[p] = in ? (is_r ? g.first_r[idx] : g.first_q[idx]) : 0u;
    }
  }
}

template <bool BIG, bool GLDS, int NT, bool I8, bool HAM, bool CEN>
__device__ __forceinline__ void dist_stage_tile_words(const GemmArgs &g, _Float16 *sAB, uint32_t row0, uint32_t col0,
                                                      const DistTileWords<BIG, NT> &w_) {
  using TC = TileCfg<BIG, NT>;
  constexpr int BM = TC::BM, BN = TC::BN, THREADS = TC::THREADS, WORD_PASSES = DistTileWords<BIG, NT>::WORD_PASSES;
  constexpr bool CENT = (I8 && !HAM) || CEN;
  constexpr int32_t NORM_SAFE = 1 << 29;
  const uint32_t tid = threadIdx.x;
  HG_DIST_TILE_WORDS
  (void)s_nq, (void)s_iq, (void)s_sq, (void)s_fq, (void)s_tq, (void)s_eq, (void)s_tot;
  const auto &w_nv = w_.w_nv, &w_info = w_.w_info, &w_slot = w_.w_slot;
  const auto &w_first = w_.w_first;
  {
    // Phase 0 (thresholded mode): dot >= j_lo * (nr + nq - dot) rewritten as dot >= c * (nr + nq) with
    // c = j_lo / (1 + j_lo) shaved by 1e-5, evaluated in f32 straight from the accumulator: one add and one compare
    // per element.  Invalid rows / columns carry +1e30, norms outside [0, 2^29] -- where the i32 denominator could
    // wrap -- carry -1e20 (phase 1 decides those).
    // i8 path: the accumulator holds G = sum a_r*a_q and
    //   dot = 4*G + 4*corrR(i,j) + 4*corrQ(i,j) - 2*e_q*S_r - 2*e_r*S_q + D*e_r*e_q,
    // so dot <= 4*G + [2|S_r| + 1016*B_i] + [2|S_q| + 508*B_j] + D  (B = the row's sum |b| over its clamped entries:
    // |corrR| <= B_i*254, |corrQ| <= B_j*127).  The bracketed per-row / per-column slacks are folded into the row and
    // column thresholds (+64 for the i32 -> f32 rounding); rows without clamped entries, the normal case, only pay
    // 2|S|.  Phase 2 evaluates the exact integer.
    const float p0_scale = (I8 || CEN) ? 0.25f : 1.f;
#pragma unroll
    for (int p = 0; p < WORD_PASSES; ++p) {
      const uint32_t t = tid + (uint32_t)p * THREADS;
      if (t >= (uint32_t)(BM + BN)) break;
      const bool is_r = t < (uint32_t)BM;
      const uint32_t idx = is_r ? row0 + t : col0 + (t - BM);
      const bool in = idx < (is_r ? g.R : g.Q);
      const int32_t nv = w_nv[p];
      s_nr[t] = nv;
      float slack = 0.f;
      if (CENT) {
        const int32_t info = w_info[p], slot = w_slot[p];
        s_ir[t] = info;
        s_sr[t] = slot;
        s_fr[t] = w_first[p];
        // (phase 2 needs the other operand's value at this entry's dimension: with the entry here that is ONE global
        // load per candidate that has one instead of two dependent ones, with nothing to hide them behind)
        uint32_t ew = 0u;
        if constexpr (I8) {
          if (((uint32_t)slot >> 14) & 255u) {
            const I8Outlier o = g.ents[w_first[p]];
            ew = (uint32_t)o.d | ((uint32_t)(uint8_t)o.b << 16);
          }
        }
        s_er[t] = ew;
        const int32_t s2 = info - (info & 1);  // 2*S
        slack = (float)(s2 < 0 ? -s2 : s2) + (is_r ? 1016.f : 508.f) * (float)(slot & 0x3fff);
        if (!is_r) slack += (float)g.hv_d + 64.f;
      }
      // (finite sentinels, so that `d - ur - tq` is never NaN: "out of range" outweighs "norm outside the safe range")
      if (HAM) s_ur[t] = !in ? 1e30f : (is_r ? 0.f : (float)g.ham_thr);  // G >= ham_thr, exact while D <= 2^24
      else {
        // (clamped: pre_b is -inf when every pair passes -- ani_th <= 0 -- and +inf when none can; left infinite, a
        // column threshold of -inf would cancel the "out of range" of a row: inf - inf, and with it the only thing
        // that keeps the rows past R out of the lane-mask path's lists)
        const float thr = fminf(fmaxf((g.pre_c * (float)nv + (is_r ? 0.f : g.pre_b) - slack) * p0_scale, -1e20f), 1e20f);
        s_ur[t] = !in ? 1e30f : ((nv < 0 || nv > NORM_SAFE) ? -1e20f : thr);
      }
    }
    if (tid < 3) s_cnt[THREADS / 64 + 1 + THREADS / 64 + tid] = 0u;  // "some candidate list is nearly full": three slots in rotation
    if (tid < (uint32_t)(THREADS / 64)) s_fill[tid] = 0u;
  }
}

// The epilogue of one tile (see the comments inside; returns when the tile's hits are in the global list).
template <bool CHUNKED, bool FULL, bool BIG, bool GLDS, int NT, bool I8, bool HAM, bool FP4, bool CEN>
__device__ __forceinline__ void dist_epilogue(const GemmArgs &g, _Float16 *sAB, uint32_t row0, uint32_t col0,
                                              dist_acc_t<I8, FP4> (&acc)[TileCfg<BIG, NT>::WTM][NT],
                                              int32_t (&iacc)[CHUNKED ? TileCfg<BIG, NT>::WTM : 1][CHUNKED ? NT : 1][4]) {
  using TC = TileCfg<BIG, NT>;
  constexpr int LROW = GLDS ? BK : LDS_ROW;  // elements per LDS row
  constexpr int BM = TC::BM, BN = TC::BN, WTM = TC::WTM, NWN = TC::NWN, THREADS = TC::THREADS;
  [[maybe_unused]] constexpr int LOADS = TC::LOADS;
  // two LDS stages of (A tile + B tile)
  [[maybe_unused]] constexpr uint32_t A_ELEMS = BM * LROW, B_ELEMS = BN * LROW;
  [[maybe_unused]] constexpr uint32_t TILE_ELEMS = A_ELEMS;           // offset of the B tile inside a stage
  [[maybe_unused]] constexpr uint32_t STAGE_ELEMS = A_ELEMS + B_ELEMS;
  [[maybe_unused]] constexpr uint32_t SROWS = THREADS / 8;            // rows covered by one staging pass
  typedef dist_acc_t<I8, FP4> acc_t;
  [[maybe_unused]] constexpr bool CENT = (I8 && !HAM) || CEN;  // the epilogue works on centred counts: info words, dot = 4 G - ...
  HG_DIST_TILE_WORDS
  // The epilogue's per-lane addressing starts again from an opaque copy of the thread index: derived from the values
  // above it is loop invariant, gets hoisted in front of the K loop and takes registers the main loop does not have
  // (the i8 kernels went through scratch: 313 spilled registers).
  uint32_t tid_opaque = threadIdx.x;
  asm volatile("" : "+v"(tid_opaque));
  {  // (closed at the end of the kernel)
  const uint32_t tid = tid_opaque, lane = tid & 63, wave = tid >> 6;
  const uint32_t wm = wave / NWN, wn = wave % NWN, fr = lane & 15, fq = lane >> 4;
  // ---- epilogue: C[row = (lane>>4)*4 + r][col = lane&15] per 16x16 tile ------------------------
  // Phase 1 (unrolled over the accumulator registers, a handful of instructions per element): one
  // multiply-compare against a conservative Jaccard bound keeps only the pairs that can reach the
  // threshold (ANI is monotone in J) and pushes them as {local i, local j, dot} into a per-wave list in
  // LDS -- idle after the K loop, whose last barrier retired all fragment reads.
  // Phase 2 (dense: one candidate per lane): exact reference arithmetic, threshold, hits compacted in
  // place, then ONE global atomic per flush.  (A per-hit atomic on the single global counter serialised
  // at ~12 ns and cost more than the GEMM: 2.30 ms vs 1.10 ms at 1.3 M hits.)
  if (HG_EXP(4)) {  // keep the accumulators alive without running the epilogue
    if (g.hit_cap == 0xFFFFFFFFu) {
      float sum = 0.f;
#pragma unroll
      for (int m = 0; m < WTM; ++m)
#pragma unroll
        for (int n = 0; n < NT; ++n) sum += acc[m][n][0] + acc[m][n][1] + acc[m][n][2] + acc[m][n][3];
      reinterpret_cast<float *>(g.hits)[tid] = sum;
    }
    return;
  }
  uint2 *cand = reinterpret_cast<uint2 *>(sAB) + wave * CAND_CAP;
  uint32_t staged = 0;  // wave-uniform
  // Phase 2 on a list: the reference's float32 ANI of every candidate (src/dist.rs:153-160), threshold; the ANI
  // overwrites the dot product in place, a miss is marked 0xFFFFFFFF (no non-negative float has that pattern), the
  // compaction happens on the way out, after the range has been reserved.  Batches of 64 candidates touch only their
  // own entries, so they are independent: the batches of ALL lists are dealt round-robin to the waves (a cluster's block
  // of hits sits in two or three waves' lists), and a wave takes them U at a time with the loads of all U in front of
  // the arithmetic.  That matters on the i8 path: G = sum a_r*a_q becomes the exact dot product through the tabulated
  // products of the clamped entries of row i / column j (~4 % of the rows have one, so nearly every batch has a lane
  // that needs them), and those are two DEPENDENT global loads -- the entry, then the other operand's value at the
  // entry's dimension: taken batch by batch they were most of phase 2's time (in-kernel stamps: 4 000 cycles per batch;
  // 6 % of the kernel at 1.3 M hits).  Here the first entry of row and column is requested for all U batches before any
  // of them is used; further entries of a row (rare) run in a loop behind a wave-uniform test.  dot = 4*sum c_r*c_q - 2*e_q*S_r - 2*e_r*S_q + D*e_r*e_q (info word = 2*S + e).
  constexpr uint32_t NW_ = THREADS / 64;
  auto phase2_group = [&](uint2 *cl, uint32_t k0, uint32_t nb, uint32_t n_list, auto uc) __attribute__((always_inline)) -> uint32_t {  // batches k0, k0 + NW_, ...; returns their hit count
    constexpr int U = decltype(uc)::value;
    uint32_t e[U], key[U], hits = 0;
    int32_t val[U];
    bool valid[U];
#pragma unroll
    for (int u = 0; u < U; ++u) {
      const uint32_t k = k0 + (uint32_t)u * NW_;
      e[u] = k * 64 + lane;
      valid[u] = k < nb && e[u] < n_list;
      const uint2 c2 = cl[valid[u] ? e[u] : 0u];  // (entry 0 exists: nb > 0)
      key[u] = c2.x, val[u] = (int32_t)c2.y;
    }
    if constexpr (HAM) {  // G = D - 2 * hamming; the pre-filter was exact
#pragma unroll
      for (int u = 0; u < U; ++u) {
        if (valid[u]) cl[e[u]].y = (uint32_t)((int32_t)g.hv_d - val[u]) >> 1;
        hits += (uint32_t)__popcll(__ballot(valid[u]));
      }
      return hits;
    } else {
      if constexpr (CEN) {  // centred f16 operands: nothing was clamped
#pragma unroll
        for (int u = 0; u < U; ++u) {
          const uint32_t li = key[u] >> 16, lj = key[u] & 0xffffu;
          const int32_t ir = s_ir[li], iq = s_iq[lj], er = ir & 1, eq = iq & 1;
          val[u] = 4 * val[u] - eq * (ir - er) - er * (iq - eq) + (er & eq) * (int32_t)g.hv_d;
        }
      }
      if constexpr (I8) {
        int32_t ir[U], iq[U], vq[U], vr[U];
        uint32_t cr[U], cq[U], f_r[U], f_q[U];
        uint32_t o_r[U], o_q[U];  // first entries: d | b << 16
#pragma unroll
        for (int u = 0; u < U; ++u) {
          const uint32_t li = key[u] >> 16, lj = key[u] & 0xffffu;
          ir[u] = s_ir[li], iq[u] = s_iq[lj];
          cr[u] = (valid[u] && !HG_EXP(64)) ? ((uint32_t)s_sr[li] >> 14) & 255u : 0u;  // count (8) | sum |b| (14)
          cq[u] = (valid[u] && !HG_EXP(64)) ? ((uint32_t)s_sq[lj] >> 14) & 255u : 0u;
          f_r[u] = s_fr[li], f_q[u] = s_fq[lj];
          o_r[u] = s_er[li], o_q[u] = s_eq[lj];
        }
        // (loads only in the lanes that have an entry -- a handful of cache lines per batch; fetched in every lane, 64
        // different lines per instruction, the group was slower than the loops it replaces)
#pragma unroll
        for (int u = 0; u < U; ++u) {
          const uint32_t gi = row0 + (key[u] >> 16), gj = col0 + (key[u] & 0xffffu);
          vq[u] = vr[u] = 0;
          if (cr[u]) vq[u] = g.raw_q[(size_t)gj * g.hv_d + (o_r[u] & 0xffffu)];
          if (cq[u]) vr[u] = reinterpret_cast<const int8_t *>(g.A)[(size_t)gi * g.ldk * 2 + (o_q[u] & 0xffffu)];  // a_i[d]: the clamped byte
        }
#pragma unroll
        for (int u = 0; u < U; ++u) {
          const int32_t er = ir[u] & 1, eq = iq[u] & 1;
          int32_t G = val[u];
          if (cr[u]) G += (int32_t)(int8_t)(o_r[u] >> 16) * ((vq[u] + eq) >> 1);      // b_i[d] * c_j[d], c = the true centred count of column j
          if (cq[u]) G += (int32_t)(int8_t)(o_q[u] >> 16) * vr[u];  // a_i[d] * b_j[d], a = the clamped byte of row i (its operand)
          if (__ballot(cr[u] > 1u || cq[u] > 1u) != 0) {  // wave-uniform, rare: further entries of a row / column
            const uint32_t gi = row0 + (key[u] >> 16), gj = col0 + (key[u] & 0xffffu);
            for (uint32_t t = 1; t < cr[u]; ++t) {
              const I8Outlier o = g.ents[f_r[u] + t];
              G += (int32_t)o.b * (((int32_t)g.raw_q[(size_t)gj * g.hv_d + o.d] + eq) >> 1);
            }
            for (uint32_t t = 1; t < cq[u]; ++t) {
              const I8Outlier o = g.ents[f_q[u] + t];
              G += (int32_t)o.b * (int32_t)reinterpret_cast<const int8_t *>(g.A)[(size_t)gi * g.ldk * 2 + o.d];
            }
          }
          val[u] = 4 * G - eq * (ir[u] - er) - er * (iq[u] - eq) + (er & eq) * (int32_t)g.hv_d;
        }
      }
#pragma unroll
      for (int u = 0; u < U; ++u) {
        if (k0 + (uint32_t)u * NW_ >= nb) break;  // wave-uniform: the group is not full
        const uint32_t li = key[u] >> 16, lj = key[u] & 0xffffu;
        const float ani = HG_EXP(128) ? (float)val[u] * 1e-9f + 99.f : ani_from_dot(val[u], s_nr[li], s_nq[lj], g.kf);
        if constexpr (FULL) {
          if (g.ani_out && valid[u]) g.ani_out[(size_t)(row0 + li) * g.Q + (col0 + lj)] = ani;
        }
        const bool hit = valid[u] && g.hit_count && ani >= g.ani_th;
        if (valid[u]) cl[e[u]].y = hit ? __float_as_uint(ani) : 0xFFFFFFFFu;
        hits += (uint32_t)__popcll(__ballot(hit));
      }
      return hits;
    }
  };
  auto write_batch = [&](const uint2 *cl, uint32_t b, uint32_t n_list, uint32_t off) __attribute__((always_inline)) -> uint32_t {  // hits written
    const uint32_t e = b + lane;
    uint2 h2 = make_uint2(0u, 0xFFFFFFFFu);
    if (e < n_list) h2 = cl[e];
    const bool hit = h2.y != 0xFFFFFFFFu;
    const unsigned long long bal = __ballot(hit);
    const uint32_t pos = off + __builtin_amdgcn_mbcnt_hi((uint32_t)(bal >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)bal, 0u));
    if (hit && pos < g.hit_cap) {
      const uint32_t li = row0 + (h2.x >> 16);
      g.hits[pos] = hg_ani_hit{g.ref_index ? g.ref_index[li] : li + g.ref_off, col0 + g.qry_off + (h2.x & 0xffffu), __uint_as_float(h2.y)};
    }
    return (uint32_t)__popcll(bal);
  };
  // Emptying the lists: ONE reservation per workgroup (same-address returning atomics serialise at ~12 ns; with noise
  // hits in every tile all 8 waves of all 256 workgroups arrive together at the end of a round, and one atomic per
  // wave kept every CU waiting ~25 us per round), and the batches of all lists dealt round-robin to the waves.
  uint32_t *s_len = s_cnt + NW_ + 1;  // the list lengths + three "some list is nearly full" flags (slot m % 3)
#ifndef HG_GROUP_U
#define HG_GROUP_U 2  /* ... in the flushes of the slab-group path, where the accumulators are live (A/B: 3, 4) */
#endif
#ifndef HG_P2_U
#define HG_P2_U 4  /* batches a wave keeps in flight in the last phase 2 of a tile (A/B: 1 = one at a time) */
#endif
  auto flush_all = [&](auto p2uc) __attribute__((always_inline)) {  // p2uc: batches a wave keeps in flight in phase 2
    constexpr uint32_t P2_U = decltype(p2uc)::value;
    if (lane == 0) s_len[wave] = staged;
#ifdef HG_DIST_STAMPS
    unsigned long long fl_t0 = 0;
    if (tid == 0 && blockIdx.x < 2048) {
      if (g_dist_tile_all[blockIdx.x][7] == 0) g_dist_tile_all[blockIdx.x][7] = __builtin_amdgcn_s_memtime();  // (append done: the first flush begins)
      fl_t0 = __builtin_amdgcn_s_memtime();
    }
#endif
    __syncthreads();
#ifdef HG_DIST_STAMPS
    if (tid == 0 && blockIdx.x < 2048)
      for (uint32_t w = 0; w < NW_; ++w) g_dist_tile_all[blockIdx.x][3] += s_len[w];
#endif
    uint2 *all = reinterpret_cast<uint2 *>(sAB);
    uint32_t nh = 0, kglob = 0;
    for (uint32_t l = 0; l < NW_; ++l) {
      const uint32_t n_list = s_len[l];
      uint2 *cl = all + l * CAND_CAP;
      const uint32_t nb = (n_list + 63) / 64, first = (wave + NW_ - (kglob % NW_)) % NW_;  // my first batch of this list
      for (uint32_t k = first; k < nb; k += P2_U * NW_) nh += phase2_group(cl, k, nb, n_list, std::integral_constant<int, (int)P2_U>{});
      kglob += nb;
    }
    HG_TSTAMP(6)
    if (lane == 0) s_cnt[wave] = nh;
    __syncthreads();
#ifdef HG_DIST_STAMPS
    unsigned long long fl_t1 = 0;
    if (tid == 0 && blockIdx.x < 2048) fl_t1 = __builtin_amdgcn_s_memtime(), g_dist_tile_all[blockIdx.x][8] += fl_t1 - fl_t0;
#endif
    if (tid == 0) {
      uint32_t total = 0;
#pragma unroll
      for (uint32_t w = 0; w < NW_; ++w) total += s_cnt[w];
      s_cnt[NW_] = total ? atomicAdd(g.hit_count, total) : 0u;
    }
    __syncthreads();
    HG_TSTAMP(7)
#ifdef HG_DIST_STAMPS
    unsigned long long fl_t2 = 0;
    if (tid == 0 && blockIdx.x < 2048) {
      fl_t2 = __builtin_amdgcn_s_memtime(), g_dist_tile_all[blockIdx.x][9] += fl_t2 - fl_t1, g_dist_tile_all[blockIdx.x][12] += 1;
      for (uint32_t w = 0; w < NW_; ++w) g_dist_tile_all[blockIdx.x][13] += s_cnt[w];
    }
#endif
    uint32_t off = s_cnt[NW_];
    for (uint32_t w = 0; w < wave; ++w) off += s_cnt[w];
    kglob = 0;
    for (uint32_t l = 0; l < NW_; ++l) {  // the same batches again: compact them into this wave's part of the range
      const uint32_t n_list = s_len[l];
      const uint2 *cl = all + l * CAND_CAP;
      const uint32_t nb = (n_list + 63) / 64, first = (wave + NW_ - (kglob % NW_)) % NW_;
      for (uint32_t k = first; k < nb; k += NW_) off += write_batch(cl, k * 64, n_list, off);
      kglob += nb;
    }
    HG_TSTAMP(8)
#ifdef HG_DIST_STAMPS
    unsigned long long fl_t3 = 0;
    if (tid == 0 && blockIdx.x < 2048) fl_t3 = __builtin_amdgcn_s_memtime(), g_dist_tile_all[blockIdx.x][10] += fl_t3 - fl_t2;
#endif
    staged = 0;
    __syncthreads();  // the lists may be refilled only after every wave has read them
#ifdef HG_DIST_STAMPS
    if (tid == 0 && blockIdx.x < 2048) g_dist_tile_all[blockIdx.x][11] += __builtin_amdgcn_s_memtime() - fl_t3;
#endif
  };
  // Phase 0: `d >= ur(row) + tq(column)` with the thresholds staged at kernel entry: the lane's NT column thresholds are
  // fetched here, the four row thresholds of a slab with one 16-byte read per slab (all 4 * WTM of them kept in
  // registers from the top push the i8 kernels into scratch).
  int32_t nqv[NT];
  float tq[NT];
#pragma unroll
  for (int n = 0; n < NT; ++n) {
    nqv[n] = s_nq[wn * (NT * 16) + n * 16 + fr];
    tq[n] = s_tq[wn * (NT * 16) + n * 16 + fr];
  }
  const float4v *s_ur4 = reinterpret_cast<const float4v *>(s_ur + wm * (WTM * 16) + fq * 4);  // slab m: s_ur4[4 m]
  HG_TSTAMP(3)
  // Two ways through the accumulators (in-kernel stamps, DESIGN.md 4.3: with one branch per element and the row words
  // read slab by slab the sweep took 20 000 cycles in a tile without a single candidate and 25 000 more in a tile with
  // the ~1 000 scattered candidates every tile of a real comparison has):
  //  * LANE MASKS (every thresholded kernel -- there a candidate is "passes phase 0"; the f16 kernels' denominator
  //    test of the slab path is only a cheaper filter in front of the exact phase 2: 0.69 -> 0.61 ms at
  //    10 000 x 10 000 without it, the windowed kernel 0.85 -> 0.78): every lane shifts the sign of `d - ur - tq` of
  //    its 4 * NT elements of a 16-row slab into one mask word per slab, no branches.  One barrier makes the waves'
  //    candidate counts known to all: a tile without candidates ends there; if no list can overflow, every wave then
  //    appends its candidates on its own -- per slab one LDS atomic per lane that has any reserves its run of the
  //    list, predicated stores fill it -- and the workgroup meets again in flush_all.
  //  * SLABS (the full-matrix mode; tiles of the windowed kernel whose candidates may overflow a list): per 16-row
  //    slab the 4 * NT compares are OR-ed on the scalar side into one wave-uniform branch; a slab with candidates
  //    takes one ballot per element, and a barrier per slab makes the decision to empty the lists uniform.
  constexpr bool LANE_MASKS = !FULL;  // (f16 operands: phase 2 is exact, the slab path's denominator test is only a cheaper filter)
  constexpr uint32_t BNC_LANE = 80, BNC_WAVE = 64 * BNC_LANE;  // bytes of a lane's / a wave's bounce buffer (append loop)
  static_assert(NT * 16 <= (int)BNC_LANE, "a lane's slab fits its bounce buffer");
  constexpr uint32_t SLAB_BITS = (1u << (4 * NT)) - 1u;
  bool by_lane = false, have_masks = false;  // workgroup-uniform
  uint32_t notpass[LANE_MASKS ? WTM : 1], lane_cands = 0, wave_cands = 0;  // (wave_cands: lane w holds wave w's count)
  if constexpr (LANE_MASKS) {
    {
      have_masks = true;
      // A tile that straddles the diagonal of a symmetric comparison (the reference's path_r == path_q case, src/dist.rs:243-265)
      // reports i < j only: its elements with global row >= global column are masked out here like elements that fail the
      // threshold, and the tile takes the same lists as every other one.  (Until round 5 such tiles -- the dense ones of a
      // database compared with itself -- went through the per-slab path below.)
      const bool on_diag = g.symmetric && row0 + g.ref_off + (uint32_t)BM - 1u >= col0 + g.qry_off;  // workgroup-uniform
      // (global row of the lane's element r = 0 of slab 0) - (global column of its n = 0 element); indices are < 2^31
      const int32_t diag0 = (int32_t)(row0 + g.ref_off + wm * (uint32_t)(WTM * 16) + fq * 4u) - (int32_t)(col0 + g.qry_off + wn * (uint32_t)(NT * 16) + fr);
      uint32_t lane_total = 0;
      dist_static_for(std::make_integer_sequence<int, WTM>{}, [&](auto mc) {
        constexpr int m = decltype(mc)::value;
        uint32_t np = 0;
        const float4v ur4 = s_ur4[4 * m];
        dist_static_for(std::make_integer_sequence<int, 4>{}, [&](auto rc) {
          constexpr int r = decltype(rc)::value;
          dist_static_for(std::make_integer_sequence<int, NT>{}, [&](auto nc) {
            constexpr int n = decltype(nc)::value;
            // (the convert as asm: written as a cast it is the same expression as in the slab path below, gets computed
            // once for both, and 160 converted accumulators stay live next to the 160 originals -- scratch)
            float d;
            if constexpr (std::is_same<acc_t, int4v>::value) asm("v_cvt_f32_i32_e32 %0, %1" : "=v"(d) : "v"(acc[m][n][r]));
            else d = acc[m][n][r];
            if constexpr (CHUNKED) {  // (+ the windows already moved into the integer accumulator; asm for the same reason)
              float di;
              asm("v_cvt_f32_i32_e32 %0, %1" : "=v"(di) : "v"(iacc[CHUNKED ? m : 0][CHUNKED ? n : 0][r]));
              d += di;
            }
            const float margin = (d - ur4[r]) - tq[n];  // (finite sentinels: never NaN)
            np = __builtin_amdgcn_alignbit(np, __float_as_uint(margin), 31);  // (np << 1) | sign: element e = r * NT + n at bit 4 NT - 1 - e
          });
        });
        if (on_diag) {
          uint32_t ex = 0;  // element e = r * NT + n at bit 4 NT - 1 - e, like np
          dist_static_for(std::make_integer_sequence<int, 4>{}, [&](auto rc) {
            dist_static_for(std::make_integer_sequence<int, NT>{}, [&](auto nc) {
              constexpr int dd = 16 * m + decltype(rc)::value - 16 * decltype(nc)::value;
              ex = (ex << 1) | (diag0 + dd >= 0 ? 1u : 0u);
            });
          });
          np |= ex;
        }
        notpass[m] = np;
        lane_total += (uint32_t)__popc(~np & SLAB_BITS);
      });
      lane_cands = lane_total;
      for (int o = 32; o > 0; o >>= 1) lane_total += __shfl_xor(lane_total, o);
      if (lane == 0) s_tot[wave] = lane_total;
      __syncthreads();
      HG_TSTAMP(9)
      wave_cands = lane < (uint32_t)(THREADS / 64) ? s_tot[lane] : 0u;
      if (__ballot(wave_cands != 0u) == 0) {  // nothing in this tile
        HG_TSTAMP(4)
        HG_TSTAMP(5)
        return;
      }
      uint32_t all_c = 0;
#pragma unroll
      for (uint32_t w = 0; w < (uint32_t)(THREADS / 64); ++w) all_c += __builtin_amdgcn_readlane(wave_cands, w);
      // (one list of all candidates below the waves' bounce buffers, see the append loop)
      by_lane = __ballot(wave_cands > CAND_CAP) == 0 && all_c * 8u <= (uint32_t)dist_lds_main_bytes<BIG, NT, GLDS>() - (THREADS / 64) * BNC_WAVE;
    }
  }
      // A lane has ~2 candidates among its 160 accumulators, at positions only it knows, and registers cannot be
      // indexed per lane: slab by slab the lane's 4 * NT accumulators bounce through LDS (NT 16-byte stores into the
      // lane's own 80 bytes -- a stride that keeps 16 lanes on 64 different banks), and a loop over the set bits of the
      // slab's mask reads the ones that pass back by address and appends them.  (The straightforward form -- one
      // predicated append per element, 160 exec-mask regions per lane -- took 8 000 cycles per tile for ~1 000
      // candidates: in-kernel stamps.)  The bounce buffers sit at the top of the stage area, the list grows from its
      // bottom; a wave's LDS operations execute in order, so no barrier is involved.
  auto append_slabs = [&](uint32_t off, uint32_t m_lo, uint32_t m_hi, uint32_t wave_u) __attribute__((always_inline)) {
    if constexpr (LANE_MASKS) {
      uint2 *const cand = reinterpret_cast<uint2 *>(sAB);
      char *const bnc = reinterpret_cast<char *>(sAB) + dist_lds_main_bytes<BIG, NT, GLDS>() - (THREADS / 64 - wave_u) * BNC_WAVE + lane * BNC_LANE;
      dist_static_for(std::make_integer_sequence<int, WTM>{}, [&](auto mc) {
        constexpr int m = decltype(mc)::value;
        if ((uint32_t)m < m_lo || (uint32_t)m >= m_hi) return;  // (constants in the one-list path)
        uint32_t rest = ~notpass[m] & SLAB_BITS;
        if (__ballot(rest != 0u) == 0) return;  // wave-uniform
#pragma unroll
        for (int n = 0; n < NT; ++n) {
          if constexpr (CHUNKED) {  // the exact dot product = last window (f32, exact) + the integer windows
            int4v v;
#pragma unroll
            for (int r = 0; r < 4; ++r) v[r] = (int32_t)acc[m][n][r] + iacc[CHUNKED ? m : 0][CHUNKED ? n : 0][r];
            *reinterpret_cast<int4v *>(bnc + n * 16) = v;
          } else {
            *reinterpret_cast<acc_t *>(bnc + n * 16) = acc[m][n];
          }
        }
        const uint32_t key0 = ((wm * (WTM * 16) + m * 16 + fq * 4) << 16) | (wn * (NT * 16) + fr);
        while (rest != 0u) {
          const uint32_t e = (uint32_t)(4 * NT - 1) - (uint32_t)__builtin_ctz(rest), r = e / (uint32_t)NT, n = e - r * (uint32_t)NT;
          rest &= rest - 1u;
          int32_t G;
          if constexpr (std::is_same<acc_t, int4v>::value || CHUNKED) G = *reinterpret_cast<const int32_t *>(bnc + n * 16 + r * 4);
          else G = (int32_t)*reinterpret_cast<const float *>(bnc + n * 16 + r * 4);
          cand[off] = make_uint2(key0 + (r << 16) + n * 16u, (uint32_t)G);
          ++off;
        }
      });
    }
  };
  if (by_lane) {
    if constexpr (LANE_MASKS) {
      // ONE list for the workgroup (the per-wave regions are contiguous): wave w's candidates start behind those of
      // the waves below it -- every wave knows all counts --, and one LDS atomic per lane reserves the run that takes
      // the lane's candidates.  Fewer half-empty batches for phase 2 than eight lists, and flush_all sees list 0 only.
      const uint32_t wave_u = __builtin_amdgcn_readfirstlane(wave);
      uint32_t off = 0, all_cands = 0;
#pragma unroll
      for (uint32_t w = 0; w < (uint32_t)(THREADS / 64); ++w) {
        const uint32_t v = __builtin_amdgcn_readlane(wave_cands, w);
        off += w < wave_u ? v : 0u;
        all_cands += v;
      }
      if (lane_cands != 0u) off += atomicAdd(&s_fill[wave], lane_cands);
      append_slabs(off, 0u, (uint32_t)WTM, wave_u);
      staged = wave_u == 0 ? all_cands : 0u;
    }
  } else if (have_masks && !CHUNKED && !HG_EXP(256)) {  // (the windowed kernel's second accumulator set leaves no registers for it)
    if constexpr (LANE_MASKS) {
      // GROUPS OF SLABS (a tile with more candidates than the one list holds: the dense diagonal blocks of a database compared
      // with itself): the same masks, the same append, the same cooperative phase 2 -- for as many whole 16-row slabs at a
      // time as the list takes (one slab of all waves always fits).  The slab path below empties the per-wave lists after
      // nearly every slab of such a tile, because ONE wave's list fills up while the others' stay empty (a 100 x 100 block
      // of hits lies in two waves' columns): 8 flushes of 3 000 candidates instead of 2-3 of 10 000, each with its
      // reservation round trip and its barriers -- a dense tile took 308 k ticks against 134 k for a normal one, and the
      // CUs that hold one ended a tile time after the others (profiles/r04_dist_defer_neutral.txt).
      constexpr uint32_t NWV = THREADS / 64;
      constexpr uint32_t TOTM_BYTES = 4u * NWV * WTM;
      constexpr uint32_t LIST_ROOM = ((uint32_t)dist_lds_main_bytes<BIG, NT, GLDS>() - NWV * BNC_WAVE - TOTM_BYTES) / 8u;
      static_assert(LIST_ROOM >= NWV * 4u * NT * 64u, "one slab of all waves fits the list");
      uint32_t *const s_totm = reinterpret_cast<uint32_t *>(reinterpret_cast<char *>(sAB) + dist_lds_main_bytes<BIG, NT, GLDS>() - NWV * BNC_WAVE - TOTM_BYTES);  // [wave][slab]
      const uint32_t wave_u = __builtin_amdgcn_readfirstlane(wave);
#pragma unroll
      for (int m = 0; m < WTM; ++m) {
        uint32_t c = (uint32_t)__popc(~notpass[m] & SLAB_BITS);
        for (int o = 32; o > 0; o >>= 1) c += __shfl_xor(c, o);
        if (lane == 0) s_totm[wave_u * WTM + m] = c;
      }
      __syncthreads();
      // (nothing of the table is kept in registers: the 160 accumulators and the masks are live across the flushes below)
      uint32_t m0 = 0;  // workgroup-uniform
      while (m0 < (uint32_t)WTM) {
        uint32_t m1 = m0, gtot = 0, goff = 0, lane_g = 0;
#pragma unroll
        for (int m = 0; m < WTM; ++m) {
          if ((uint32_t)m != m1 || (uint32_t)m < m0) continue;  // uniform
          uint32_t t = 0, bl = 0;  // candidates of slab m in all waves / in the waves below this one
          for (uint32_t w = 0; w < NWV; ++w) {
            const uint32_t v = s_totm[w * WTM + m];
            t += v, bl += w < wave_u ? v : 0u;
          }
          if (gtot + t > LIST_ROOM) continue;
          gtot += t, goff += bl, lane_g += (uint32_t)__popc(~notpass[m] & SLAB_BITS), m1 = (uint32_t)m + 1u;
        }
        if (gtot != 0u) {
          if (lane == 0) s_fill[wave] = 0u;  // (this wave's LDS operations execute in order: the reset is in front of its lanes' atomics)
          uint32_t off = goff;
          if (lane_g != 0u) off += atomicAdd(&s_fill[wave], lane_g);
          append_slabs(off, m0, m1, wave_u);
          staged = wave_u == 0 ? gtot : 0u;
          flush_all(std::integral_constant<uint32_t, HG_GROUP_U>{});
        }
        m0 = m1;
      }
      HG_TSTAMP(4)
      HG_TSTAMP(5)
      return;
    }
  } else {
  // (compile-time m, r, n: the accumulator registers must be indexed statically whatever the optimiser thinks of the
  // size of the unrolled body -- a loop it declines to unroll sends all 160 accumulators through scratch)
  dist_static_for(std::make_integer_sequence<int, WTM>{}, [&](auto mc) {
    constexpr int m = decltype(mc)::value;
    const float4v ur4 = s_ur4[4 * m];
    auto passes = [&](auto rc, auto nc) __attribute__((always_inline)) -> bool {
      constexpr int r = decltype(rc)::value, n = decltype(nc)::value;
      if constexpr (FULL) return true;
      else if constexpr (HAM) return (int32_t)acc[m][n][r] >= g.ham_thr;
      else {
        float d = (float)acc[m][n][r];
        if (CHUNKED) d += (float)iacc[CHUNKED ? m : 0][CHUNKED ? n : 0][r];
        return d >= ur4[r] + tq[n];
      }
    };
    unsigned long long slab = FULL ? ~0ull : 0ull;
    if constexpr (!FULL) {
      dist_static_for(std::make_integer_sequence<int, 4>{}, [&](auto rc) {
        dist_static_for(std::make_integer_sequence<int, NT>{}, [&](auto nc) { slab |= __ballot(passes(rc, nc)); });
      });
    }
    if (slab != 0) {  // wave-uniform
      dist_static_for(std::make_integer_sequence<int, 4>{}, [&](auto rc) {
        constexpr int r = decltype(rc)::value;
        const uint32_t li = wm * (WTM * 16) + m * 16 + fq * 4 + r, i = row0 + li;
        const bool iok = i < g.R;
        dist_static_for(std::make_integer_sequence<int, NT>{}, [&](auto nc) {
          constexpr int n = decltype(nc)::value;
          const bool pass = passes(rc, nc);
          if (__ballot(pass) == 0) return;  // wave-uniform
          const uint32_t lj = wn * (NT * 16) + n * 16 + fr, j = col0 + lj;
          int32_t dot = (int32_t)acc[m][n][r];
          if (CHUNKED) dot = (int32_t)((uint32_t)dot + (uint32_t)iacc[CHUNKED ? m : 0][CHUNKED ? n : 0][r]);
          bool live = pass && iok && j < g.Q && !(g.symmetric && i + g.ref_off >= j + g.qry_off);
          if (!FULL && !I8 && !CEN) {  // (centred operands: the list carries the raw G, phase 2 forms the exact dot product)
            const int32_t den = (int32_t)((uint32_t)s_nr[li] + (uint32_t)nqv[n] - (uint32_t)dot);
            live = live && (den <= 0 || (float)dot >= g.j_lo * (float)den);
          }
          const unsigned long long bal = __ballot(live);
          if (live) {
            const uint32_t pos =
                staged + __builtin_amdgcn_mbcnt_hi((uint32_t)(bal >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)bal, 0u));
            cand[pos] = make_uint2((li << 16) | lj, (uint32_t)dot);
          }
          staged += (uint32_t)__popcll(bal);
        });
      });
    }
    // At most 4 * NT * 64 candidates per m and wave.  A list that might overflow in the next m (dense blocks of hits
    // only) is emptied by the WHOLE workgroup: the decision is made uniform through LDS, one barrier per m.
    if constexpr (m + 1 < WTM) {
      // Three flag slots in rotation: slot m % 3 is raised before this m's barrier and read after it; the slot of
      // m + 2 is cleared here, between barrier m and barrier m + 1 -- every wave read it (as slot m - 1) before it
      // arrived at barrier m, and nobody raises it before barrier m + 1.  (With ONE slot a fast wave could raise the
      // flag for m + 1 before a slow one had read it for m: the two would then disagree about the flush.)
      if (lane == 0 && staged > CAND_CAP - 4 * NT * 64) s_len[NW_ + m % 3] = 1u;
      __syncthreads();
      const bool any_full = s_len[NW_ + m % 3] != 0u;
      if (wave == 0 && lane == 0) s_len[NW_ + (m + 2) % 3] = 0u;
      if (any_full) flush_all(std::integral_constant<uint32_t, (HG_P2_U < 2 ? HG_P2_U : 2)>{});  // (the accumulators are live: two batches in flight)
    }
  });
  }
  HG_TSTAMP(4)
  flush_all(std::integral_constant<uint32_t, HG_P2_U>{});  // end of the tile
  HG_TSTAMP(5)
  }
}

}  // namespace
