// hg_dist_gemm.h -- geometry, launch arguments and development switches of dist_mfma_kernel (private to hg_dist_kernels.hip).
#pragma once
#include <vector>

#include "hg_dist_prep.h"

namespace {

// ---- MFMA GEMM + ANI ------------------------------------------------------------------------------
constexpr int BK = 64;
constexpr int LDS_ROW = BK + 8;  // f16 elements per LDS row: 144 B => conflict-free ds_read_b128
constexpr uint32_t ST = 8;       // super-tile edge, in tiles
// workgroups of a launch: whole super-tiles
static inline uint32_t dist_grid(uint32_t tiles_m, uint32_t tiles_n) {
  return ((tiles_m + ST - 1) / ST) * ((tiles_n + ST - 1) / ST) * ST * ST;
}
// Slot -> tile table of a launch (see the kernel's tile order): the tiles that have work -- inside the matrix, not entirely
// on / below the diagonal of a symmetric comparison -- in the XCD-aware super-tile walk, cut into 8 runs of equal length
// (+-1), run x walked by the workgroups b with b % 8 == x; with `diag` the tiles that straddle the diagonal (two per tile
// row) come first, dealt round the XCDs.  Returns the grid size and sets g.tile_tab / g.diag_first; on any failure (or
// for shapes beyond 65 535 tiles a side, or under the "legacy" order hook) the kernel's own blockIdx mapping stays.
static uint32_t dist_tile_table(hg_ctx *c, struct GemmArgs &g, uint32_t bm, uint32_t bn, bool diag);
// Tile geometries (waves are 2 (M) x NWN (N), each wave owns WTM x NT MFMA tiles of 16 x 16):
//   small: 128 x 128, 4 waves, 72 KiB LDS, 2 workgroups / CU  -- small problems, little padding
//   big  : 256 x 256, 8 waves, 144 KiB LDS, 1 workgroup / CU  -- half the LDS and L2 bytes per flop
//   wide : 256 x 320 (NT = 5, LDS-DMA only): chosen when it divides the tile grid into fewer rounds over the
//          CUs (10 000 x 10 000: 1 280 tiles = 5.0 rounds of 256 instead of 1 600 = 6.25 -> 7)
// (Measured and removed again, see DESIGN.md 4.3 / 4.4 and the history of this file: a four-wave 128 x 128 per-wave shape
// with AGPR-pinned accumulators, bit-stream operands expanded by the workgroup, a register-staged 256 x 256 variant,
// the DMA burst spread over all waves -- also with the two waves of a SIMD half a phase apart --, raised priority for the
// loader waves, hand-written DMA issue with one M0 write per four pieces, and a ping-pong main loop in which the two
// waves of a SIMD alternate between a 40-MFMA burst and fragment reads + DMA over a four-slice ring.)
template <bool BIG, int NT = 4>
struct TileCfg {
  static constexpr int WTM = BIG ? 8 : 4;   // 16-row MFMA tiles per wave in M
  static constexpr int NWN = BIG ? 4 : 2;   // waves in N
  static constexpr int BM = 2 * WTM * 16, BN = NWN * NT * 16;  // NT = 16-column MFMA tiles per wave in N
  static constexpr int THREADS = 2 * NWN * 64;
  static constexpr int LOADS = BM * BK * 2 / 16 / THREADS;    // 16-byte pieces per thread, A operand
  static constexpr int LOADS_B = BN * BK * 2 / 16 / THREADS;  // ... B operand
};

// Dynamic LDS of dist_mfma_kernel: [ two operand stages | the epilogue's per-wave candidate lists, whichever is larger ]
// followed by the tile's row / column words (norms, pre-filter thresholds, i8 info words; staged at kernel entry, so
// they live beside the stages: 161 472 B for the 256 x 320 tile).
constexpr uint32_t CAND_CAP = 2048;  // candidates per wave list: 16 KiB
template <bool BIG, int NT, bool GLDS>
constexpr size_t dist_lds_main_bytes() {
  using TC = TileCfg<BIG, NT>;
  const size_t stages = (size_t)2 * (TC::BM + TC::BN) * (GLDS ? BK : LDS_ROW) * sizeof(_Float16);
  const size_t lists = (size_t)(TC::THREADS / 64) * CAND_CAP * 8;
  return stages > lists ? stages : lists;
}
template <bool BIG, int NT, bool GLDS>
constexpr size_t dist_lds_bytes() {
  return dist_lds_main_bytes<BIG, NT, GLDS>() + (size_t)6 * (TileCfg<BIG, NT>::BM + TileCfg<BIG, NT>::BN) * 4 + 192;
}

struct GemmArgs {
  const _Float16 *A;  // Rp x Kp (ref)
  const _Float16 *B;  // Qp x Kp (query)
  const int32_t *nr, *nq;
  uint32_t R, Q, Kp;
  uint32_t ldk;          // row pitch of A and B in elements (Kp + pad, see hg_run_dist)
  uint32_t chunk_steps;  // K-steps (of BK) per exact f32 accumulation window
  float kf;
  float *ani_out;
  hg_ani_hit *hits;
  uint32_t *hit_count;
  uint32_t hit_cap;
  float ani_th;
  float j_lo;  // conservative Jaccard bound: dot < j_lo * den  =>  ANI < ani_th for sure
  float pre_c, pre_b;  // phase-0 form of the same bound: dot < pre_c * (nr + nq) + pre_b  =>  rejected
  int symmetric;
  uint32_t ref_off, qry_off;  // global index of row 0 / column 0 (a block of a larger matrix): hits and the i < j test use them
  uint32_t tiles_m, tiles_n;  // tile grid
  const uint32_t *verdict;    // speculative launch: runs only if v_lo <= verdict[0] <= v_hi (see decide_kernel);
  uint32_t v_lo, v_hi;        // with chunk_from_verdict the window length (K-steps) is verdict[1]
  uint32_t chunk_from_verdict;
  const uint32_t *veto;       // f16 kernels queued behind an i8 attempt: return at once if *veto == 1 (i8 path valid)
  // i8 operand path (I8 instantiations): row / column info words 2*S + e, control words of the i8 prepass
  const int32_t *info_r, *info_q;
  const int32_t *slot_r, *slot_q;  // entries (8 bits) | sum |b| (14) per row / column, 0 = none
  const uint32_t *first_r, *first_q;  // ... and where the row's / column's entries start in `ents`
  uint32_t ent_cap;                // capacity of `ents`
  uint32_t *i8verdict;             // [0] <- 1 when the i8 attempt is valid, [1] <- K-steps (written by workgroup 0: the
                                   // host's read-back and the veto word of the f16 kernels queued behind)
  const I8Outlier *ents;           // clamped entries sorted by (side, row, dim)
  const int16_t *raw_q;            // the original i16 query matrix (rows of hv_d): c_j[d] for the reference rows' clamped entries
                                   // (a_i[d] for the query columns' entries is the reference's byte operand itself: A)
  const uint32_t *ref_index;       // optional: global index of reference row i (a gathered block whose rows are not one
                                   // contiguous range of the global enumeration); nullptr: ref_off + i
  const uint32_t *i8ctrl;          // [0] entries reserved, [1] flags of the prepass
  uint32_t hv_d, same_set;
  uint32_t diag_first;             // leading workgroup slots that take the tiles on the diagonal (0: plain order)
  const uint32_t *tile_tab;        // slot -> tile (tm | tn << 16, ~0u: no tile) built by the host (dist_tile_table); nullptr:
                                   // the workgroup derives its tile from blockIdx as described at the top of the kernel
  int32_t ham_thr;                 // HAM: candidates are G >= ham_thr
};

// development builds only (-DHG_DIST_EXPERIMENT=<bits>, results are wrong): timing with parts of the kernel
// compiled out -- 1 no in-loop DMA, 2 no fragment reads / MFMAs, 4 no epilogue, 8 reads but no MFMAs, 16 no in-loop barrier, 32 fragments read in the first step only,
// 64 no outlier corrections in phase 2, 512 every tile streams the operand rows of tile (0, 0) -- and only their first 2 KiB, over and over: all L2 hits (what the L2 misses cost)
#ifdef HG_DIST_EXPERIMENT
#define HG_EXP(bit) ((HG_DIST_EXPERIMENT & (bit)) != 0)
#else
#define HG_EXP(bit) false
#endif

// Development builds only (-DHG_DIST_STAMPS): s_memtime stamps of the main loop's phases, written by lane 0 of waves 0
// (a loader) and 5 (no loads) of the first 16 workgroups for K-steps 8..15 into g_dist_stamps[wg][wave][step][point]
// (tools/dist_stamps.py reads them back through hg_debug_dist_stamps).  Points: 0 step top, 1 fragments of the last
// phase requested (before the lgkmcnt wait), 2 before the barrier, 3 behind the barrier, 4 behind the DMA issue, 5 step end.
#ifdef HG_DIST_STAMPS
__device__ unsigned long long g_dist_stamps[16][2][8][6];
#define HG_STAMP(pt)                                                                                       \
  if (lane == 0 && (wave == 0 || wave == 5) && blockIdx.x < 16 && ks >= 8 && ks < 16)                      \
    g_dist_stamps[blockIdx.x][wave == 5][ks - 8][pt] = __builtin_amdgcn_s_memtime();
// ... and of the tile as a whole: lane 0 of every wave of workgroups 512..527 (the third round of tiles) into
// g_dist_tile_stamps[wg][wave][point].  Points: 0 kernel entry, 1 first stage landed, 2 main loop done, 3 norms staged,
// 4 accumulator sweep done, 5 lists emptied (tile done); inside the last flush_all: 6 candidates evaluated, 7 range
// reserved, 8 hits written.
__device__ unsigned long long g_dist_tile_stamps[16][8][10];
__device__ unsigned long long g_dist_tile_real[2048][2];  // s_memrealtime (100 MHz) at points 1 and 2: with the stamps of
                                                          // g_dist_tile_stamps' wave 0 this gives the shader clock of the main loop
// per workgroup: 0 entry, 1 main loop done, 2 tile done, 3 candidates evaluated, 4 XCC id | HW_ID << 8, 5 norms staged (point 3),
// 6 phase-0 masks built and the waves' counts known (point 9), 7 candidates appended / sweep done (point 4; the first time),
// and summed over the tile's flushes: 8 flush start -> candidates evaluated (phase 2), 9 -> range reserved, 10 -> hits
// written, 11 -> lists free again (the closing barrier); 12 flushes, 13 hits
__device__ unsigned long long g_dist_tile_all[2048][16];
#define HG_TSTAMP(pt)                                                                                      \
  if ((threadIdx.x & 63) == 0 && blockIdx.x >= 512 && blockIdx.x < 528 && (pt) < 9)                        \
    g_dist_tile_stamps[blockIdx.x - 512][threadIdx.x >> 6][pt] = __builtin_amdgcn_s_memtime();            \
  if (threadIdx.x == 0 && blockIdx.x < 2048 && ((pt) == 0 || (pt) == 2 || (pt) == 5))                      \
    g_dist_tile_all[blockIdx.x][(pt) == 0 ? 0 : ((pt) == 2 ? 1 : 2)] = __builtin_amdgcn_s_memtime();       \
  if (threadIdx.x == 0 && blockIdx.x < 2048 && ((pt) == 3 || (pt) == 9))                                   \
    g_dist_tile_all[blockIdx.x][(pt) == 3 ? 5 : 6] = __builtin_amdgcn_s_memtime();                         \
  if (threadIdx.x == 0 && blockIdx.x < 2048 && (pt) == 4 && g_dist_tile_all[blockIdx.x][7] == 0)           \
    g_dist_tile_all[blockIdx.x][7] = __builtin_amdgcn_s_memtime();                                         \
  if (threadIdx.x == 0 && blockIdx.x < 2048 && ((pt) == 1 || (pt) == 2))                                   \
    g_dist_tile_real[blockIdx.x][(pt)-1] = __builtin_amdgcn_s_memrealtime();
#else
#define HG_STAMP(pt)
#define HG_TSTAMP(pt)
#endif

// accumulator of one 16 x 16 MFMA tile: i8 operands accumulate in exact i32, everything else (f16, e2m1) in f32
template <bool I8, bool FP4>
using dist_acc_t = typename std::conditional<I8 && !FP4, int4v, float4v>::type;

// The tile's row / column words live behind the operand stages (dist_lds_bytes): norms, i8 info / slot / first-entry
// words, the phase-0 thresholds and the epilogue's counters.  Declares the pointers (needs BIG, NT, GLDS, BM, BN, THREADS
// and sAB in scope): written by dist_stage_tile_words at kernel entry, read by the epilogue.
#define HG_DIST_TILE_WORDS                                                                                                             \
  int32_t *s_nr = reinterpret_cast<int32_t *>(reinterpret_cast<char *>(sAB) + dist_lds_main_bytes<BIG, NT, GLDS>()), *s_nq = s_nr + BM; \
  int32_t *s_ir = s_nq + BN, *s_iq = s_ir + BM; \
  int32_t *s_sr = s_iq + BN, *s_sq = s_sr + BM; \
  uint32_t *s_fr = reinterpret_cast<uint32_t *>(s_sq + BN), *s_fq = s_fr + BM; \
  float *s_ur = reinterpret_cast<float *>(s_fq + BN), *s_tq = s_ur + BM; \
  uint32_t *s_er = reinterpret_cast<uint32_t *>(s_tq + BN), *s_eq = s_er + BM; \
  uint32_t *s_cnt = s_eq + BN; \
  uint32_t *s_tot = s_cnt + 2 * (THREADS / 64) + 4, *s_fill = s_tot + THREADS / 64;

}  // namespace
