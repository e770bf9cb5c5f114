// hg_dist_kernels.hip -- all-pairs hypervector ANI on gfx950.
//
// Replaces dist::compute_hv_ani / compute_pairwise_ani (src/dist.rs:139-161,231-294): for every
// (ref, query) pair   dot = sum_d r[d]*q[d]  (i32),  J = dot / (|r|^2 + |q|^2 - dot),
// ANI = 1 + ln(2J/(1+J))/k, clamped, x100 -- float32 in the reference's operation order.
//
// The contraction is a dense R x Q x D GEMM and runs on the matrix cores.  The HV entries are
// small integers (|x| ~ sqrt(n_hashes)), so they are converted once to f16 (exact for
// |x| <= 2048) and multiplied with v_mfma_f32_16x16x32_f16; products are exact in f32, and the
// f32 accumulator is exact as long as sum |r||q| over the accumulated K range stays below 2^24.
// The prepass measures a guaranteed Cauchy-Schwarz bound for that sum: normally the whole-row bound
// shows that ONE accumulation window covers K (verdict formed on the device, the GEMM queued
// behind it speculatively); otherwise a second prepass measures the bound per K-chunk and the
// kernel moves the accumulator into i32 registers at chunk boundaries chosen from it.  If no chunk
// size is safe (or |x| > 2048) the always-exact integer VALU kernel is used instead.  Either
// way the dot product equals the reference's i32 value bit for bit; only logf differs from
// glibc by <= 1 ulp.  Sketch hypervectors (hv = 2 * count - n) normally take the centred i8 operand path instead
// (hg_dist_prep.h: prep_i8_kernel), twice the K per instruction on v_mfma_i32_16x16x64_i8.
//
// Where what lives (all private to this translation unit):
//   hg_dist_common.h      vector typedefs, the reference's float32 ANI (ani_from_dot)
//   hg_dist_prep.h        operand prepasses and their on-device exactness verdicts
//   hg_dist_gemm.h        tile geometry, LDS layout, GemmArgs, development switches (-DHG_DIST_EXPERIMENT, -DHG_DIST_STAMPS)
//   hg_dist_tile_order.h  the host-built workgroup slot -> tile table
//   hg_dist_mainloop.h    dist_main_loop: the K loop (LDS-DMA staging, fragment reads, MFMAs)
//   hg_dist_epilogue.h    tile words staged at kernel entry, pre-filter, candidate lists, exact ANI, hit list
//   this file             the kernel skeleton, the integer fallback, the bit -> operand expanders, the launch logic
#include <algorithm>
#include <atomic>
#include <cmath>
#include <cstdlib>
#include <cstring>
#include <type_traits>
#include <utility>
#include <vector>

#include "hg_dist_tile_order.h"
#include "hg_dist_mainloop.h"
#include "hg_dist_epilogue.h"

namespace {

// FULL: every ANI is evaluated and stored (parity / small problems).  Otherwise only pairs that can
// reach ani_th are evaluated: one multiply-compare rejects the rest (ANI is monotone in the Jaccard
// index), the exact reference arithmetic decides the survivors.
// GLDS (big geometry only): operand tiles go HBM -> LDS by LDS-DMA (global_load_lds, 16 B per lane, no
// VGPR staging and no ds_write pass).  The DMA writes each wave-instruction's 1 KiB linearly, so the
// LDS image is unpadded [row][8 chunks of 16 B] and bank conflicts are removed by an XOR swizzle of the
// chunk index with (row >> 1) & 7 -- applied to the per-lane SOURCE address when loading and to the
// fragment address when reading (same involution on both sides).
// HAM (with I8): the operands are +-1 expanded from bit-packed hypervectors, G = D - 2*hamming; the epilogue keeps
// G >= ham_thr and reports {ref, qry, (D - G) / 2} -- the bit-packed search on the matrix pipe.  Two operand formats:
//   bytes   (+-1 as i8, v_mfma_i32_16x16x64_i8: K = 64 dims per instruction), or
//   FP4     (+-1.0 as e2m1 nibbles 0x2 / 0xA, v_mfma_scale_f32_16x16x128_f8f6f4 with unit block scales: K = 128 dims
//           per instruction at the same cycles, exact in the f32 accumulator while D <= 2^24).  Either way a lane's
//           fragment is 16 bytes and a K-step is 128 bytes per row, so staging, swizzle and fragment addressing are
//           shared; the order of the dims inside a fragment is irrelevant as long as both operands use the same one
//           (every product is +-1 and they are all summed).
// CEN (f16 operands): the operands are the CENTRED counts c = (x + e) >> 1 of sketch hypervectors (hv = 2 * count - n,
// src/hd.rs:29,84-87) as f16 -- half the magnitude of x, so the Cauchy-Schwarz bound that proves the f32 accumulator exact
// over ALL of K holds up to ~16 000 hashes per sketch at D = 4096 instead of ~4 000 (sum c^2 = n D / 4) -- and the
// epilogue recovers dot = 4 G - 2 e_q S_r - 2 e_r S_q + D e_r e_q from the row / column info words like the i8 path,
// without clamped entries.  Sketches too large for byte operands (beyond ~6 000 hashes) take this kernel instead of the
// windowed one with its i32 side accumulators and 256 x 192 tiles.
template <bool CHUNKED, bool FULL, bool BIG, bool GLDS = false, int NT = 4, bool I8 = false, bool HAM = false, bool FP4 = false,
          bool CEN = false>
__global__ __launch_bounds__((TileCfg<BIG, 4>::THREADS)) void dist_mfma_kernel(GemmArgs g) {
  using TC = TileCfg<BIG, NT>;
  static_assert(!CEN || (!I8 && GLDS && !CHUNKED && !FULL), "centred f16 operands: thresholded whole-K LDS-DMA geometries");
  static_assert(!I8 || (GLDS && !CHUNKED && !FULL), "the i8 operand path exists for the thresholded LDS-DMA geometries");
  static_assert(!HAM || I8, "the Hamming epilogue rides on the i8 operand path");
  static_assert(!FP4 || HAM, "e2m1 operands exist for the Hamming search only");
  HG_TSTAMP(0)
#ifdef HG_DIST_STAMPS
  if (threadIdx.x == 0 && blockIdx.x < 2048)
    g_dist_tile_all[blockIdx.x][3] = 0, g_dist_tile_all[blockIdx.x][1] = 0, g_dist_tile_all[blockIdx.x][2] = 0,
    g_dist_tile_all[blockIdx.x][5] = 0, g_dist_tile_all[blockIdx.x][6] = 0, g_dist_tile_all[blockIdx.x][7] = 0,
    g_dist_tile_all[blockIdx.x][8] = 0, g_dist_tile_all[blockIdx.x][9] = 0, g_dist_tile_all[blockIdx.x][10] = 0,
    g_dist_tile_all[blockIdx.x][11] = 0, g_dist_tile_all[blockIdx.x][12] = 0, g_dist_tile_all[blockIdx.x][13] = 0,
    g_dist_tile_all[blockIdx.x][4] = __builtin_amdgcn_s_getreg(20 | (0 << 6) | (3 << 11)) |  // hwreg(HW_REG_XCC_ID, 0, 4)
                                     ((unsigned long long)__builtin_amdgcn_s_getreg(4 | (0 << 6) | (31 << 11)) << 8);  // HW_REG_HW_ID
#endif
  if (g.veto && (CEN ? g.veto[0] == 1u : g.veto[0] != 0u)) return;  // uniform: a kernel queued before this one did the work (1: the i8 one, 2: the centred f16 one)
  if (I8 && !HAM) {
    const bool ok = i8_attempt_valid(g.i8ctrl, g.ent_cap);
    if (blockIdx.x == 0 && threadIdx.x == 0) g.i8verdict[0] = ok ? 1u : 0u, g.i8verdict[1] = g.Kp / BK;
    if (!ok) return;
  }
  if (g.verdict) {  // uniform
    const uint32_t code = g.verdict[0];
    if (code < g.v_lo || code > g.v_hi) return;
    if (CHUNKED && g.chunk_from_verdict) g.chunk_steps = g.verdict[1];
  }
  static_assert(!GLDS || BIG, "LDS-DMA variant exists for the 256 x 256 geometry only");
  constexpr int BM = TC::BM, BN = TC::BN, WTM = TC::WTM;
  extern __shared__ __attribute__((aligned(16))) _Float16 sAB[];

  // XCD-aware tile order (MI355X guide T1): workgroups b and b + 8 run on the same XCD and share its 4 MiB L2, so every
  // XCD gets a contiguous run of tiles (bijective remap) and walks 8 x 8 super-tiles inside it: the 32 workgroups resident
  // on an XCD cover 4 x 8 tiles -- 4 A row-blocks and 8 B row-blocks through that L2 instead of 32 different B blocks
  // (measured in round 1: 7.0 GB of L2 misses per 10k x 10k launch with plain row-major order) --, and the next 4 x 8
  // tiles reuse the same 8 B blocks.  (Round 3 tried 4 x 8 super-tiles dealt round-robin to the XCDs, to spread the
  // expensive diagonal tiles of a self-comparison evenly: the same time on i8 operands, but the L2 hit rate fell from
  // 68 % to 64 %, and from 63 % to 49 % on f16 operands -- without the shared B blocks between consecutive groups.  The
  // diagonal is dealt with below.)
  // A database compared with itself in file order has its hits on the diagonal, and a tile with 20 000 candidates spends
  // twice as long in its epilogue as in its K loop: with five tiles per CU the launch ends when the last such tile does.
  // The host may therefore put the tiles that straddle the diagonal in front (g.diag_first workgroup slots, two per tile
  // row, a multiple of 8 so that the XCD of the remaining workgroups is unchanged): longest jobs first.
  uint32_t tm, tn;
  if (g.tile_tab) {
    // the host's table: exactly the tiles that have work, every XCD the same number of them (+-1) in the order described
    // above, the diagonal ones in front.  Slots that return at once -- the super-tile grid's padding, the diagonal tiles'
    // places in the walk, the lower triangle of a symmetric comparison -- made some CUs run six tiles and others four
    // where five each were due: the hardware deals workgroup i to XCD i % 8 whatever it turns out to do.
    const uint32_t t = g.tile_tab[blockIdx.x];
    if (t == ~0u) return;
    tm = t & 0xFFFFu, tn = t >> 16;
  } else if (blockIdx.x < g.diag_first) {
    // slot s: tile row s % (diag_first / 2), its first (s < diag_first / 2) or second diagonal tile -- diag_first / 2 is a
    // multiple of 8, so the rows' first tiles, the dense ones, go round the XCDs (with two adjacent slots per row they
    // all fell to the even XCDs: 223 k against 126 k candidates per XCD)
    const uint32_t half = g.diag_first >> 1, second = blockIdx.x >= half ? 1u : 0u;
    tm = blockIdx.x - second * half;
    if (tm >= g.tiles_m) return;
    const uint32_t tn0 = tm * BM / BN, tn1 = (tm * BM + BM - 1) / BN;
    tn = second ? tn1 : tn0;
    if ((second && tn1 == tn0) || tn >= g.tiles_n) return;
  } else {
    const uint32_t b = blockIdx.x - g.diag_first, nwg = gridDim.x - g.diag_first;
    const uint32_t q = nwg / 8, r = nwg % 8, xcd = b % 8;
    const uint32_t bid = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + b / 8;
    const uint32_t sup_n = (g.tiles_n + ST - 1) / ST;
    const uint32_t sup = bid / (ST * ST), within = bid % (ST * ST);
    tm = (sup / sup_n) * ST + within / ST, tn = (sup % sup_n) * ST + within % ST;
    if (tm >= g.tiles_m || tn >= g.tiles_n) return;  // padding of the super-tile grid
    if (g.diag_first && (tn == tm * BM / BN || tn == (tm * BM + BM - 1) / BN)) return;  // ran in front
  }
  const uint32_t row0 = tm * BM, col0 = tn * BN;
  if (g.symmetric && row0 + g.ref_off >= col0 + g.qry_off + BN) return;  // tile entirely on/below the diagonal

  dist_acc_t<I8, FP4> acc[WTM][NT];
  int32_t iacc[CHUNKED ? WTM : 1][CHUNKED ? NT : 1][4];
#pragma unroll
  for (int m = 0; m < WTM; ++m)
#pragma unroll
    for (int n = 0; n < NT; ++n) acc[m][n] = dist_acc_t<I8, FP4>{};
  if (CHUNKED) {
#pragma unroll
    for (int m = 0; m < WTM; ++m)
#pragma unroll
      for (int n = 0; n < NT; ++n)
#pragma unroll
        for (int r = 0; r < 4; ++r) iacc[CHUNKED ? m : 0][CHUNKED ? n : 0][r] = 0;
  }
  // the three parts (hg_dist_epilogue.h, hg_dist_mainloop.h): the tile's row / column words are requested first, the main
  // loop publishes them with its first operand stage, the epilogue turns the accumulators into hits
  DistTileWords<BIG, NT> words;
  dist_load_tile_words<BIG, NT, I8, HAM, CEN>(g, row0, col0, words);
  dist_main_loop<CHUNKED, BIG, GLDS, NT, I8, FP4>(g, sAB, row0, col0, acc, iacc,
                                                  [&]() { dist_stage_tile_words<BIG, GLDS, NT, I8, HAM, CEN>(g, sAB, row0, col0, words); });
  dist_epilogue<CHUNKED, FULL, BIG, GLDS, NT, I8, HAM, FP4, CEN>(g, sAB, row0, col0, acc, iacc);
}

// ---- always-exact integer fallback -------------------------------------------------------------------
// 16 x 16 outputs per workgroup, K staged through LDS in slices of 128 dims.
constexpr int FB_T = 16, FB_K = 128;
__global__ __launch_bounds__(FB_T *FB_T) void dist_int_kernel(const int16_t *__restrict__ ref,
                                                               const int16_t *__restrict__ qry, hg_dist_args a,
                                                               float kf) {
  __shared__ int16_t sR[FB_T][FB_K + 2];
  __shared__ int16_t sQ[FB_T][FB_K + 2];
  const uint32_t tx = threadIdx.x & 15, ty = threadIdx.x >> 4;
  const uint32_t i0 = blockIdx.y * FB_T, j0 = blockIdx.x * FB_T;
  if (a.symmetric && i0 + a.ref_off >= j0 + a.qry_off + FB_T) return;
  uint32_t acc = 0;
  for (uint32_t k0 = 0; k0 < a.hv_d; k0 += FB_K) {
    __syncthreads();
    for (uint32_t e = threadIdx.x; e < FB_T * FB_K; e += FB_T * FB_T) {
      const uint32_t r = e / FB_K, k = e % FB_K;
      const bool kin = k0 + k < a.hv_d;
      sR[r][k] = (kin && i0 + r < a.R) ? ref[(size_t)(i0 + r) * a.hv_d + k0 + k] : (int16_t)0;
      sQ[r][k] = (kin && j0 + r < a.Q) ? qry[(size_t)(j0 + r) * a.hv_d + k0 + k] : (int16_t)0;
    }
    __syncthreads();
#pragma unroll 8
    for (uint32_t k = 0; k < FB_K; ++k) acc += (uint32_t)((int32_t)sR[ty][k] * (int32_t)sQ[tx][k]);
  }
  const uint32_t i = i0 + ty, j = j0 + tx;
  if (i >= a.R || j >= a.Q) return;
  if (a.symmetric && i + a.ref_off >= j + a.qry_off) return;
  const float ani = ani_from_dot((int32_t)acc, a.ref_n2[i], a.qry_n2[j], kf);
  if (a.ani_out) a.ani_out[(size_t)i * a.Q + j] = ani;
  if (a.hit_count && ani >= a.ani_th) {
    const uint32_t idx = atomicAdd(a.hit_count, 1u);
    if (idx < a.hit_cap) a.hits[idx] = hg_ani_hit{i + a.ref_off, j + a.qry_off, ani};
  }
}


// ---- a handful of sketches against a database (hyper-gen search / dist with one or a few genomes on one side) ----------
// With <= 16 rows on one side the tiles of the matrix-pipe kernels are 3-6 % used and the other side's operand prepass alone
// costs more than reading it: here the small side sits in LDS as it is (int16), every wave streams rows of the large side
// once (16 bytes per lane and load, a row of 4 096 dimensions = eight loads in flight), forms the exact int32 dot products
// with v_dot2_i32_i16 (wrapping like the reference's i32 sum, src/dist.rs:147-151), reduces them across the wave and lets
// lane s finish pair (row, s): ANI, threshold, one aggregated append per wave.  100 000 x 1 / 10 / 16 at D = 4 096:
// 0.53 / 0.54 / 0.54 -> 0.155 / 0.23 / 0.48 ms per call (the 820 MB of rows stream in 0.15).  SWAP: the small side is the REFERENCE side (rows of the matrix).
constexpr uint32_t SK_T = 512, SK_MAX = 16, SK_C = 8;  // threads, rows of the small side, 16-byte pieces per lane and pass
typedef short short2s __attribute__((ext_vector_type(2)));
template <bool SWAP>
__global__ __launch_bounds__(SK_T) void dist_skinny_kernel(const int16_t *__restrict__ big, const int32_t *__restrict__ big_n2,
                                                           uint32_t n_big, const int16_t *__restrict__ sml,
                                                           const int32_t *__restrict__ sml_n2, uint32_t n_sml, hg_dist_args a,
                                                           float kf) {
  extern __shared__ __attribute__((aligned(16))) uint4 s_sml[];  // n_sml rows of hv_d / 8 pieces
  const uint32_t pieces = a.hv_d / 8, lane = threadIdx.x & 63u, wave = threadIdx.x >> 6;
  for (uint32_t i = threadIdx.x; i < n_sml * pieces; i += SK_T) s_sml[i] = reinterpret_cast<const uint4 *>(sml)[i];
  __syncthreads();
  // Two rows of the large side per wave and step: every 16-byte read of the small side serves both (with sixteen small rows
  // the LDS reads, 128 KB per large row, were what the kernel waited for: 0.69 ms for 100 000 x 16, slower than the tiles)
  const uint32_t waves = gridDim.x * (SK_T / 64);
  for (uint32_t row0 = 2 * (blockIdx.x * (SK_T / 64) + wave); row0 < n_big; row0 += 2 * waves) {  // wave-uniform
    const bool two = row0 + 1 < n_big;
    const uint4 *__restrict__ src0 = reinterpret_cast<const uint4 *>(big + (size_t)row0 * a.hv_d);
    const uint4 *__restrict__ src1 = reinterpret_cast<const uint4 *>(big + (size_t)(two ? row0 + 1 : row0) * a.hv_d);
    int32_t acc0[SK_MAX], acc1[SK_MAX];
#pragma unroll
    for (uint32_t q = 0; q < SK_MAX; ++q) acc0[q] = 0, acc1[q] = 0;
    for (uint32_t p0 = 0; p0 < pieces; p0 += 64 * SK_C) {
      uint4 v0[SK_C], v1[SK_C];
#pragma unroll
      for (uint32_t c = 0; c < SK_C; ++c) {
        const uint32_t p = p0 + c * 64 + lane;
        v0[c] = p < pieces ? src0[p] : make_uint4(0, 0, 0, 0);
        v1[c] = p < pieces ? src1[p] : make_uint4(0, 0, 0, 0);
      }
#pragma unroll
      for (uint32_t q = 0; q < SK_MAX; ++q) {
        if (q < n_sml) {  // wave-uniform
          int32_t t0 = acc0[q], t1 = acc1[q];
#pragma unroll
          for (uint32_t c = 0; c < SK_C; ++c) {
            const uint32_t p = p0 + c * 64 + lane;
            if (p0 + c * 64 < pieces) {  // wave-uniform
              const uint4 w = s_sml[q * pieces + (p < pieces ? p : 0)];  // (a lane past the row's end multiplies zeros)
              t0 = __builtin_amdgcn_sdot2(__builtin_bit_cast(short2s, v0[c].x), __builtin_bit_cast(short2s, w.x), t0, false);
              t0 = __builtin_amdgcn_sdot2(__builtin_bit_cast(short2s, v0[c].y), __builtin_bit_cast(short2s, w.y), t0, false);
              t0 = __builtin_amdgcn_sdot2(__builtin_bit_cast(short2s, v0[c].z), __builtin_bit_cast(short2s, w.z), t0, false);
              t0 = __builtin_amdgcn_sdot2(__builtin_bit_cast(short2s, v0[c].w), __builtin_bit_cast(short2s, w.w), t0, false);
              t1 = __builtin_amdgcn_sdot2(__builtin_bit_cast(short2s, v1[c].x), __builtin_bit_cast(short2s, w.x), t1, false);
              t1 = __builtin_amdgcn_sdot2(__builtin_bit_cast(short2s, v1[c].y), __builtin_bit_cast(short2s, w.y), t1, false);
              t1 = __builtin_amdgcn_sdot2(__builtin_bit_cast(short2s, v1[c].z), __builtin_bit_cast(short2s, w.z), t1, false);
              t1 = __builtin_amdgcn_sdot2(__builtin_bit_cast(short2s, v1[c].w), __builtin_bit_cast(short2s, w.w), t1, false);
            }
          }
          acc0[q] = t0, acc1[q] = t1;
        }
      }
    }
    // wave totals; lane q keeps row 0's total of small row q, lane 32 + q row 1's
    int32_t mine = 0;
#pragma unroll
    for (uint32_t q = 0; q < SK_MAX; ++q) {
      if (q < n_sml) {  // wave-uniform
        int32_t t0 = acc0[q], t1 = acc1[q];
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) t0 += __shfl_xor(t0, o), t1 += __shfl_xor(t1, o);
        if (lane == q) mine = t0;
        if (lane == 32 + q) mine = t1;
      }
    }
    bool hit = false;
    float ani = 0.0f;
    const uint32_t sq = lane & 31u, row = row0 + (lane >> 5);  // this lane's small row and large row
    const uint32_t i = SWAP ? sq : row, j = SWAP ? row : sq;   // (reference row, query column) of its pair
    if (sq < n_sml && (lane < 32 || two) && !(a.symmetric && i + a.ref_off >= j + a.qry_off)) {
      ani = SWAP ? ani_from_dot(mine, sml_n2[sq], big_n2[row], kf) : ani_from_dot(mine, big_n2[row], sml_n2[sq], kf);
      hit = ani >= a.ani_th;
    }
    const unsigned long long bal = __ballot(hit);
    if (bal) {  // wave-uniform
      uint32_t base = 0;
      if (lane == 0) base = atomicAdd(a.hit_count, (uint32_t)__popcll(bal));
      base = __builtin_amdgcn_readfirstlane(base);
      const uint32_t idx = base + (uint32_t)__popcll(bal & ((1ull << lane) - 1ull));
      if (hit && idx < a.hit_cap) a.hits[idx] = hg_ani_hit{i + a.ref_off, j + a.qry_off, ani};
    }
  }
}

}  // namespace

// the instantiation's name as a profiler prints it (hg_ctx_last_kernel: bench.py matches it against the kernel names in
// the committed rocprofv3 summaries before it quotes their counters)
template <bool CHUNKED, bool FULL, bool BIG, bool GLDS = false, int NT = 4, bool I8 = false, bool HAM = false, bool FP4 = false,
          bool CEN = false>
static std::string dist_kernel_name() {
  auto b = [](bool x) { return x ? "true" : "false"; };
  return std::string("dist_mfma_kernel<") + b(CHUNKED) + ", " + b(FULL) + ", " + b(BIG) + ", " + b(GLDS) + ", " +
         std::to_string(NT) + ", " + b(I8) + ", " + b(HAM) + ", " + b(FP4) + ", " + b(CEN) + ">";
}
#define HG_DIST_K(...) &dist_mfma_kernel<__VA_ARGS__>, dist_kernel_name<__VA_ARGS__>()

// ---- bit-packed Hamming search on the matrix pipe ---------------------------------------------------------------
// bits -> +-1 bytes (bit 1 -> +1, bit 0 -> -1), one lane per 32-bit word: per nibble the four bits are spread to the
// low bit of four bytes by one multiply ((x * 0x00204081) & 0x01010101: the partial products never collide) and
// turned into 0x01 / 0xFF by a byte permute from a two-entry table.
__global__ __launch_bounds__(256) void expand_bits_kernel(const uint32_t *__restrict__ bits, uint32_t rows, uint32_t words,
                                                          uint32_t ldk8, int8_t *__restrict__ out) {
  const uint32_t row = blockIdx.y;
  for (uint32_t w = blockIdx.x * blockDim.x + threadIdx.x; w < words; w += gridDim.x * blockDim.x) {
    const uint32_t v = bits[(size_t)row * words + w];
    uint32_t d[8];
#pragma unroll
    for (int i = 0; i < 8; ++i) {
      const uint32_t m = (((v >> (4 * i)) & 0xFu) * 0x00204081u) & 0x01010101u;
      d[i] = __builtin_amdgcn_perm(0u, 0x000001FFu, m);  // selector byte 0 -> 0xFF (-1), 1 -> 0x01 (+1)
    }
    uint4 *dst = reinterpret_cast<uint4 *>(out + (size_t)row * ldk8 + (size_t)w * 32);
    dst[0] = make_uint4(d[0], d[1], d[2], d[3]);
    dst[1] = make_uint4(d[4], d[5], d[6], d[7]);
  }
}

// bits -> e2m1 nibbles (bit 1 -> +1.0 = 0x2, bit 0 -> -1.0 = 0xA), one lane per 32-bit word = 16 bytes of output.  The
// eight bits of a byte are spread to bit 0 of eight nibbles by three shift-or-mask steps; the code is
// 0x2 | (!bit << 3).  Words past the row's end (K is padded to whole 128-byte K-steps) become zero nibbles: +0.0
// contributes nothing to G.  Which dim lands in which nibble does not matter as long as both operands are expanded by
// this kernel (see dist_mfma_kernel).
__global__ __launch_bounds__(256) void expand_bits_fp4_kernel(const uint32_t *__restrict__ bits, uint32_t rows, uint32_t words,
                                                              uint32_t groups, uint32_t ldk4, uint8_t *__restrict__ out) {
  const uint32_t row = blockIdx.y;
  for (uint32_t w = blockIdx.x * blockDim.x + threadIdx.x; w < groups; w += gridDim.x * blockDim.x) {
    uint32_t d[4] = {0u, 0u, 0u, 0u};
    if (w < words) {
      const uint32_t v = ~bits[(size_t)row * words + w];
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        uint32_t y = (v >> (8 * i)) & 0xFFu;
        y = (y | (y << 12)) & 0x000F000Fu;
        y = (y | (y << 6)) & 0x03030303u;
        y = (y | (y << 3)) & 0x11111111u;
        d[i] = (y << 3) | 0x22222222u;
      }
    }
    *reinterpret_cast<uint4 *>(out + (size_t)row * ldk4 + (size_t)w * 16) = make_uint4(d[0], d[1], d[2], d[3]);
  }
}

// path: 1 = +-1 byte operands on v_mfma_i32_16x16x64_i8 (hv_d a multiple of 128), 2 = e2m1 operands on
// v_mfma_scale_f32_16x16x128_f8f6f4 (any hv_d: rows are whole 32-bit words; pad bits count like in the popcount kernel)
hg_status hg_run_hamming_mfma(hg_ctx *c, const uint32_t *d_ref_bits, uint32_t R, const uint32_t *d_qry_bits, uint32_t Q,
                              uint32_t hv_d, uint32_t max_dist, hg_ham_hit *d_hits, uint32_t *d_count, uint32_t cap,
                              uint32_t ref_off, uint32_t qry_off, int path) {
  static_assert(sizeof(hg_ham_hit) == sizeof(hg_ani_hit), "the GEMM epilogue writes 12-byte records");
  const bool fp4 = path == 2;
  const uint32_t words = (hv_d + 31) / 32, dims = words * 32;  // G = dims - 2 * popcount(xor of whole words)
  if (!fp4 && hv_d % 128) return hg_fail(c, HG_ERR_INVALID, "byte operands need hv_d % 128 == 0");
  // row bytes: one per dim (bytes) or half of one (e2m1), padded to whole 128-byte K-steps, pitch + 128 B (see hg_run_dist)
  const uint32_t kbytes = fp4 ? (dims / 2 + 127) / 128 * 128 : hv_d, ldkb = kbytes + 128;
  auto padded = [](uint32_t n) { return std::max({(n + 255) / 256 * 256, (n + 319) / 320 * 320, (n + 191) / 192 * 192}); };
  const uint32_t Rp = padded(R), Qp = padded(Q);
  hg_status s;
  if ((s = hg_ensure(c, c->w_i8a, (size_t)Rp * ldkb)) != HG_OK) return s;
  if ((s = hg_ensure(c, c->w_i8b, (size_t)Qp * ldkb)) != HG_OK) return s;
  auto *a8 = static_cast<uint8_t *>(c->w_i8a.p), *b8 = static_cast<uint8_t *>(c->w_i8b.p);
  c->i8_pad[0].ptr = c->i8_pad[1].ptr = nullptr;  // (the dist path's zero rows in these buffers are overwritten below)
  c->i8_sig_ref = c->i8_sig_qry = nullptr;  // the dist path's operand copies are gone
  if (Rp > R) HG_HIP(c, hipMemsetAsync(a8 + (size_t)R * ldkb, 0, (size_t)(Rp - R) * ldkb, c->stream));
  if (Qp > Q) HG_HIP(c, hipMemsetAsync(b8 + (size_t)Q * ldkb, 0, (size_t)(Qp - Q) * ldkb, c->stream));
  {
    hg_timed tp(c, HG_T_DIST_PREP);
    const uint32_t groups = kbytes / 16;
    const unsigned gx = fp4 ? (groups + 255) / 256 : (words + 255) / 256;
    auto expand = [&](const uint32_t *bits, uint32_t n, uint8_t *out) -> hipError_t {
      for (uint32_t r0 = 0; r0 < n; r0 += 65535) {
        const uint32_t m = std::min<uint32_t>(65535, n - r0);
        if (fp4)
          hipLaunchKernelGGL(expand_bits_fp4_kernel, dim3(gx, m), dim3(256), 0, c->stream, bits + (size_t)r0 * words, m, words,
                             groups, ldkb, out + (size_t)r0 * ldkb);
        else
          hipLaunchKernelGGL(expand_bits_kernel, dim3(gx, m), dim3(256), 0, c->stream, bits + (size_t)r0 * words, m, words, ldkb,
                             reinterpret_cast<int8_t *>(out + (size_t)r0 * ldkb));
        const hipError_t e = hipGetLastError();
        if (e != hipSuccess) return e;
      }
      return hipSuccess;
    };
    HG_HIP(c, expand(d_ref_bits, R, a8));
    HG_HIP(c, expand(d_qry_bits, Q, b8));
  }
  GemmArgs g{};
  g.A = reinterpret_cast<const _Float16 *>(a8), g.B = reinterpret_cast<const _Float16 *>(b8);
  g.R = R, g.Q = Q, g.Kp = kbytes / 2, g.ldk = ldkb / 2, g.chunk_steps = ~0u;  // in two-byte units (a K-step is 128 bytes)
  g.hits = reinterpret_cast<hg_ani_hit *>(d_hits), g.hit_count = d_count, g.hit_cap = cap;
  g.ref_off = ref_off, g.qry_off = qry_off, g.hv_d = dims;
  // dist <= max  <=>  G = D - 2*dist >= D - 2*max  (max >= D: everything is a hit)
  g.ham_thr = max_dist >= dims ? -(int32_t)dims - 1 : (int32_t)dims - 2 * (int32_t)max_dist;
  int nt = 4;
  {
    const uint64_t tm = (R + 255) / 256, ncu = (uint64_t)std::max(c->n_cu, 1);
    const uint64_t r4 = (tm * ((Q + 255) / 256) + ncu - 1) / ncu, r5 = (tm * ((Q + 319) / 320) + ncu - 1) / ncu;
    // a 256 x 320 tile is priced at 1.25 x 0.9 of a 256 x 256 one (13 % fewer fragment bytes per MFMA; 50 000 x 10 000 x
    // 16384 on byte operands: 5.9 ms against 6.55 at equal padded area)
    if (r5 * 9 < r4 * 8) nt = 5;
    if (c->dbg_dist_tile == "big") nt = 4;
    else if (c->dbg_dist_tile == "wide") nt = 5;
  }
  g.tiles_m = (R + 255) / 256, g.tiles_n = (Q + (uint32_t)nt * 64 - 1) / ((uint32_t)nt * 64);
  const uint32_t n_tiles = dist_tile_table(c, g, 256, (uint32_t)nt * 64, false);
  const size_t lds = nt == 5 ? dist_lds_bytes<true, 5, true>() : dist_lds_bytes<true, 4, true>();
  auto launch = [&](auto kern, const std::string &name, int threads) -> hipError_t {
    c->last_kernel[HG_T_DIST] = name;
    const void *fp = reinterpret_cast<const void *>(kern);
    if (std::find(c->lds_attr_done.begin(), c->lds_attr_done.end(), fp) == c->lds_attr_done.end()) {
      const hipError_t e = hipFuncSetAttribute(fp, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
      if (e != hipSuccess) return e;
      c->lds_attr_done.push_back(fp);
    }
    hipLaunchKernelGGL(kern, dim3(n_tiles), dim3(threads), lds, c->stream, g);
    return hipGetLastError();
  };
  hg_timed tg(c, HG_T_DIST);
  hipError_t le;
  if (fp4 && nt == 5) le = launch(HG_DIST_K(false, false, true, true, 5, true, true, true), TileCfg<true, 5>::THREADS);
  else if (fp4) le = launch(HG_DIST_K(false, false, true, true, 4, true, true, true), TileCfg<true, 4>::THREADS);
  else if (nt == 5) le = launch(HG_DIST_K(false, false, true, true, 5, true, true), TileCfg<true, 5>::THREADS);
  else le = launch(HG_DIST_K(false, false, true, true, 4, true, true), TileCfg<true, 4>::THREADS);
  HG_HIP(c, le);
  return HG_OK;
}

// ANI >= th  <=>  J >= x/(2-x) with x = exp(k*(th/100-1)).  Returned with a relative safety
// margin far above the float32 rounding of the device-side test (3 roundings of 2^-24), so a pair
// rejected by `dot < j_lo*den` can never reach the threshold; survivors are re-tested exactly.
static float jaccard_lower_bound(float ani_th, uint32_t ksize) {
  if (!(ani_th > 0.0f)) return -INFINITY;   // everything is reported
  if (ani_th > 100.0f) return INFINITY;     // nothing can be (ANI is clamped to 100)
  const double x = std::exp((double)ksize * ((double)ani_th / 100.0 - 1.0));
  const double j = x / (2.0 - x);
  return (float)(j * (1.0 - 1e-4));
}

hg_status hg_run_dist(hg_ctx *c, const hg_dist_args &a, uint32_t *d_verdict, int *speculated) {
  if (speculated) *speculated = -1;
  const uint32_t Kp = (a.hv_d + BK - 1) / BK * BK;
  // Row pitch of the f16 copies: Kp + 64 elements (+128 B).  With a power-of-two pitch (8 KiB at
  // D = 4096) every workgroup reads the same 128-byte column offset of 256 different rows at the same
  // moment, i.e. one L2 / Infinity-Cache channel; the odd 128-byte skew spreads rows over channels.
  const uint32_t ldk = Kp + 64;
  // padded row counts cover every tile geometry: 128- and 256-row tiles, 320-wide tiles and the 192-wide tiles of
  // the windowed (CHUNKED) big geometry -- the LDS-DMA reads whole tiles, rows past R / Q must exist and be zero
  auto padded = [](uint32_t n) {
    return std::max(std::max((n + 255) / 256 * 256, (n + 319) / 320 * 320), (n + 191) / 192 * 192);
  };
  const uint32_t Rp = padded(a.R), Qp = padded(a.Q);
  // ops_given: the reference side arrives as byte operands + control records prepared where the rows live
  // (hg_dist_prep_ops_dev on the owning GPUs, gathered by the caller): no reference prepass here, and no f16 fallback --
  // there are no i16 reference rows to fall back on; a veto comes back to the caller as HG_ERR_INEXACT
  const bool ops_given = a.ref_ops != nullptr;
  const bool same = !ops_given && (a.ref_hv == a.qry_hv) && (a.R == a.Q);
  hg_status s;
  // ---- a handful of rows on one side: the streaming kernel (no operand prepass, no tiles)
  {
    const bool q_small = a.Q <= a.R;
    const uint32_t n_sml = q_small ? a.Q : a.R, n_big = q_small ? a.R : a.Q;
    const size_t lds = (size_t)n_sml * a.hv_d * sizeof(int16_t);
    const bool aligned = ((reinterpret_cast<uintptr_t>(a.ref_hv) | reinterpret_cast<uintptr_t>(a.qry_hv)) & 15) == 0;
    // (a count-only call -- no hit buffer, capacity 0: the trailing blocks of a comparison whose buffer is full -- streams too:
    // the kernel writes a hit only below hit_cap)
    if (!ops_given && !a.ani_out && (a.hits || a.hit_cap == 0) && a.hit_count && n_sml >= 1 && n_sml <= SK_MAX && a.hv_d % 8 == 0 && aligned &&
        lds <= 128 * 1024 && c->dbg_dist_path.empty() && c->dbg_dist_tile.empty()) {
      static std::atomic<uint64_t> done{0};
      int dev = 0;
      HG_HIP(c, hipGetDevice(&dev));
      if (dev >= 0 && dev < 64 && !((done.load() >> dev) & 1)) {
        HG_HIP(c, hipFuncSetAttribute(reinterpret_cast<const void *>(&dist_skinny_kernel<false>), hipFuncAttributeMaxDynamicSharedMemorySize, 128 * 1024));
        HG_HIP(c, hipFuncSetAttribute(reinterpret_cast<const void *>(&dist_skinny_kernel<true>), hipFuncAttributeMaxDynamicSharedMemorySize, 128 * 1024));
        done.fetch_or(1ull << dev);
      }
      const float kf = (float)a.ksize;
      const uint32_t rows_per_wg = 2 * (SK_T / 64);
      const uint32_t grid = std::min<uint32_t>((n_big + rows_per_wg - 1) / rows_per_wg, 256u * 8u);
      hg_timed tm(c, HG_T_DIST);
      c->last_dist_path = 2;  // exact integer dot products
      c->last_kernel[HG_T_DIST] = q_small ? "dist_skinny_kernel<false>" : "dist_skinny_kernel<true>";
      if (q_small)
        hipLaunchKernelGGL(dist_skinny_kernel<false>, dim3(grid), dim3(SK_T), lds, c->stream, a.ref_hv, a.ref_n2, n_big, a.qry_hv, a.qry_n2, n_sml, a, kf);
      else
        hipLaunchKernelGGL(dist_skinny_kernel<true>, dim3(grid), dim3(SK_T), lds, c->stream, a.qry_hv, a.qry_n2, n_big, a.ref_hv, a.ref_n2, n_sml, a, kf);
      HG_HIP(c, hipGetLastError());
      return HG_OK;
    }
  }
  // ---- i8 operand attempt (thresholded, large problems): queued first; every f16 kernel below carries its verdict
  // word as a veto and returns at once when the i8 kernels did the work.  After a failed attempt the next few calls
  // go straight to f16 (large sketches never qualify; probing them every time would cost ~50 us per call).
  const uint32_t *veto = nullptr;
  const bool i8_possible = d_verdict && !a.ani_out && a.hits && a.hv_d <= 8192 && a.hv_d % 8 == 0 &&
                           ((uint64_t)a.R + a.Q) * I8_ROW_SLOTS < ((uint64_t)1 << 31);  // (32-bit entry indices)
  if (ops_given && !i8_possible) return hg_fail(c, HG_ERR_UNSUPPORTED, "prepared operands: thresholded calls with hv_d <= 8192, hv_d % 8 == 0 only");
  const bool want_i8 = ops_given || (c->dbg_dist_path != "f16" && i8_possible &&
                                     ((uint64_t)a.R * a.Q >= (uint64_t)256 * 256 * 256 || c->dbg_dist_path == "i8") &&
                                     (c->i8_skip == 0 || c->dbg_dist_path == "i8"));
  if (!want_i8 && c->i8_skip) --c->i8_skip;
  if (want_i8) {
    const uint32_t kp8 = (a.hv_d + 127) / 128 * 128, ldk8 = kp8 + 128;
    if (!ops_given && (s = hg_ensure(c, c->w_i8a, (size_t)Rp * ldk8)) != HG_OK) return s;
    if (!same && (s = hg_ensure(c, c->w_i8b, (size_t)Qp * ldk8)) != HG_OK) return s;
    auto al = [](size_t x) { return (x + 255) & ~(size_t)255; };
    // misc block: info, slot and first-entry words per row / column, the list of clamped entries: I8_ROW_SLOTS per row.
    // (Where the path stops paying is the per-candidate corrections and the wider pre-filter slack, not the list:
    // 10 000 x 10 000, 1.29 M hits, GEMM + prepass -- 3 333 hashes 0.40 + 0.03 ms (f16 operands 0.66 + 0.05), 4 500:
    // 0.42 + 0.04 (0.77 + 0.04), 5 500: 0.50 + 0.04 (0.77 + 0.04), 6 000: 0.57 + 0.04 (0.75 + 0.05); beyond ~6 300
    // hashes some row of 10 000 overflows its slots and the call runs on f16 operands.)
    const uint32_t ent_cap = (uint32_t)(((uint64_t)a.R + (same ? 0 : a.Q)) * I8_ROW_SLOTS);
    const size_t o_iq = al((size_t)a.R * 4), o_sr = o_iq + al((size_t)a.Q * 4), o_sq = o_sr + al((size_t)a.R * 4);
    const size_t o_fr = o_sq + al((size_t)a.Q * 4), o_fq = o_fr + al((size_t)a.R * 4), o_list = o_fq + al((size_t)a.Q * 4);
    if ((s = hg_ensure(c, c->w_i8misc, o_list + al((size_t)ent_cap * sizeof(I8Outlier)) + 256)) != HG_OK) return s;
    // (prepared operands: the caller's buffer holds hg_dist_ops_padded_rows(R) rows; the rows behind R are zeroed below)
    auto *a8 = ops_given ? reinterpret_cast<int8_t *>(const_cast<uint8_t *>(a.ref_ops)) : static_cast<int8_t *>(c->w_i8a.p);
    auto *b8 = same ? a8 : static_cast<int8_t *>(c->w_i8b.p);
    auto *mb = static_cast<uint8_t *>(c->w_i8misc.p);
    auto *info_r = reinterpret_cast<int32_t *>(mb), *info_q = same ? info_r : reinterpret_cast<int32_t *>(mb + o_iq);
    auto *slot_r = reinterpret_cast<int32_t *>(mb + o_sr), *slot_q = same ? slot_r : reinterpret_cast<int32_t *>(mb + o_sq);
    auto *first_r = reinterpret_cast<uint32_t *>(mb + o_fr), *first_q = same ? first_r : reinterpret_cast<uint32_t *>(mb + o_fq);
    auto *list = reinterpret_cast<I8Outlier *>(mb + o_list);
    uint32_t *ctrl = d_verdict + 3;  // words 4.. of the caller's result block (zeroed by the caller, read back with the hit count)
    // The rows behind R / Q (the LDS-DMA reads whole tiles) must be zero.  The context's own operand copies keep them from
    // call to call -- the prepass writes rows below R only --, so a repeat of the same geometry needs no memset (one or two
    // 1 MB commands in front of the prepass of every call otherwise); a caller's buffer (prepared operands) is zeroed always.
    auto pad_rows = [&](int side, int8_t *base, uint32_t n, uint32_t np, bool own) -> hipError_t {
      hg_ctx::I8Pad &k = c->i8_pad[side];
      if (own && k.ptr == base && k.rows == n && k.padded == np && k.pitch == ldk8) return hipSuccess;
      k.ptr = nullptr;
      if (np > n) {
        const hipError_t e = hipMemsetAsync(base + (size_t)n * ldk8, 0, (size_t)(np - n) * ldk8, c->stream);
        if (e != hipSuccess) return e;
      }
      if (own) k.ptr = base, k.rows = n, k.padded = np, k.pitch = ldk8;
      return hipSuccess;
    };
    HG_HIP(c, pad_rows(0, a8, a.R, Rp, !ops_given));
    if (!same) HG_HIP(c, pad_rows(1, b8, a.Q, Qp, true));
    {
      hg_timed tmp(c, HG_T_DIST_PREP);
      if (ops_given)
        hipLaunchKernelGGL(unpack_meta_kernel, dim3((a.R * I8_ROW_SLOTS + 255) / 256), dim3(256), 0, c->stream,
                           static_cast<const I8RowMeta *>(a.ref_meta), a.R, info_r, slot_r, first_r, list, a.ref_flags, a.n_flags, ctrl);
      else
        hipLaunchKernelGGL(prep_i8_kernel, dim3((a.R + 3) / 4), dim3(256), 0, c->stream, a.ref_hv, a.R, a.hv_d, kp8, ldk8, a8,
                           info_r, slot_r, first_r, list, 0u, ctrl, 0u, static_cast<I8RowMeta *>(nullptr));
      HG_HIP(c, hipGetLastError());
      if (!same) {
        hipLaunchKernelGGL(prep_i8_kernel, dim3((a.Q + 3) / 4), dim3(256), 0, c->stream, a.qry_hv, a.Q, a.hv_d, kp8, ldk8, b8,
                           info_q, slot_q, first_q, list, a.R * I8_ROW_SLOTS, ctrl, 1u, static_cast<I8RowMeta *>(nullptr));
        HG_HIP(c, hipGetLastError());
      }
    }
    GemmArgs g{};
    g.A = reinterpret_cast<const _Float16 *>(a8), g.B = reinterpret_cast<const _Float16 *>(b8);
    g.nr = a.ref_n2, g.nq = a.qry_n2, g.R = a.R, g.Q = a.Q;
    g.Kp = kp8 / 2, g.ldk = ldk8 / 2;  // in two-byte units, like the f16 operands (a K-step is 128 bytes either way)
    g.chunk_steps = ~0u, g.kf = (float)a.ksize;
    g.hits = a.hits, g.hit_count = a.hit_count, g.hit_cap = a.hit_cap, g.ani_th = a.ani_th;
    g.symmetric = a.symmetric, g.ref_off = a.ref_off, g.qry_off = a.qry_off;
    g.j_lo = jaccard_lower_bound(a.ani_th, a.ksize);
    if (g.j_lo == -INFINITY) g.pre_c = 0.f, g.pre_b = -INFINITY;
    else if (g.j_lo == INFINITY) g.pre_c = 0.f, g.pre_b = INFINITY;
    else g.pre_c = (float)((double)g.j_lo / (1.0 + (double)g.j_lo) * (1.0 - 1e-5)), g.pre_b = 0.f;
    g.info_r = info_r, g.info_q = info_q, g.slot_r = slot_r, g.slot_q = slot_q, g.ents = list;
    g.first_r = first_r, g.first_q = first_q, g.ent_cap = ent_cap, g.i8verdict = ctrl + 4;
    g.raw_q = a.qry_hv, g.ref_index = a.ref_index, g.i8ctrl = ctrl, g.hv_d = a.hv_d, g.same_set = same ? 1u : 0u;
    int nt = 4;
    {
      const uint64_t tm = (a.R + 255) / 256, ncu = (uint64_t)std::max(c->n_cu, 1);
      const uint64_t r4 = (tm * ((a.Q + 255) / 256) + ncu - 1) / ncu, r5 = (tm * ((a.Q + 319) / 320) + ncu - 1) / ncu;
      if (r5 * 9 < r4 * 8) nt = 5;  // (256 x 320 costs 1.25 x 0.9 of 256 x 256, see hg_run_hamming_mfma)
      if (c->dbg_dist_tile == "big") nt = 4;
      else if (c->dbg_dist_tile == "wide") nt = 5;
    }
    g.tiles_m = (a.R + 255) / 256, g.tiles_n = (a.Q + (uint32_t)nt * 64 - 1) / ((uint32_t)nt * 64);
    // (the same matrix on both sides at the same global offset: hits cluster on the diagonal -- those tiles first)
    const uint32_t n_tiles = dist_tile_table(c, g, 256, (uint32_t)nt * 64, same && a.ref_off == a.qry_off && c->dbg_dist_order != "plain");
    const size_t lds = nt == 5 ? dist_lds_bytes<true, 5, true>() : dist_lds_bytes<true, 4, true>();
    const void *fp = nt == 5 ? reinterpret_cast<const void *>(&dist_mfma_kernel<false, false, true, true, 5, true>)
                             : reinterpret_cast<const void *>(&dist_mfma_kernel<false, false, true, true, 4, true>);
    if (std::find(c->lds_attr_done.begin(), c->lds_attr_done.end(), fp) == c->lds_attr_done.end()) {
      HG_HIP(c, hipFuncSetAttribute(fp, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
      c->lds_attr_done.push_back(fp);
    }
    {
      hg_timed tmg(c, HG_T_DIST, HG_T_DIST_PREP);
      c->last_kernel_i8 = nt == 5 ? dist_kernel_name<false, false, true, true, 5, true>() : dist_kernel_name<false, false, true, true, 4, true>();
      if (nt == 5)
        hipLaunchKernelGGL((dist_mfma_kernel<false, false, true, true, 5, true>), dim3(n_tiles), dim3(TileCfg<true, 5>::THREADS), lds,
                           c->stream, g);
      else
        hipLaunchKernelGGL((dist_mfma_kernel<false, false, true, true, 4, true>), dim3(n_tiles), dim3(TileCfg<true, 4>::THREADS), lds,
                           c->stream, g);
      HG_HIP(c, hipGetLastError());
    }
    veto = ctrl + 4;
    if (ops_given) {  // (nothing to fall back on: the caller reads the verdict)
      if (speculated) *speculated = -3;
      return HG_OK;
    }
    // The previous call on exactly these operands took the i8 path: the f16 fallback chain (five launches that would
    // all return at once) is not queued again.  Should the verdict come back negative after all, the caller reruns
    // the statistics-driven f16 schedule (*speculated == -2).
    if (c->i8_sig_ref == a.ref_hv && c->i8_sig_qry == a.qry_hv && c->i8_sig_r == a.R && c->i8_sig_q == a.Q && c->i8_sig_d == a.hv_d) {
      if (speculated) *speculated = -2;
      return HG_OK;
    }
  }
  if ((s = hg_ensure(c, c->w_f16a, (size_t)Rp * ldk * 2)) != HG_OK) return s;
  if (!same && (s = hg_ensure(c, c->w_f16b, (size_t)Qp * ldk * 2)) != HG_OK) return s;
  if ((s = hg_ensure(c, c->w_stats, 256 + 2 * PREP_SLOT_VALS * PREP_SLOTS * sizeof(unsigned long long))) != HG_OK) return s;
  auto *fa = static_cast<_Float16 *>(c->w_f16a.p);
  auto *fb = same ? fa : static_cast<_Float16 *>(c->w_f16b.p);
  // ---- centred f16 operands (thresholded, large problems; sketches that byte operands cannot hold): the counts
  // c = (x + e) >> 1 as f16 are exact in ONE f32 window up to ~16 000 hashes per sketch at D = 4096 (the raw values: ~4 000),
  // so these sketches take the whole-K kernel (256 x 256 / 320 tiles, no i32 side accumulators) instead of the windowed one.
  // Queued behind the i8 attempt and in front of the raw-value chain; `mark` (the i8 verdict word) says who did the work.
  const bool want_cen = d_verdict && !a.ani_out && a.hits && a.hv_d % 8 == 0 && c->dbg_dist_path != "f16" &&
                        ((uint64_t)a.R * a.Q >= (uint64_t)256 * 256 * 256 || c->dbg_dist_path == "cen");
  if (want_cen) {
    uint32_t *mark = d_verdict + 7;  // = ctrl[4], the i8 attempt's verdict word (zero when no attempt was queued)
    auto al = [](size_t x) { return (x + 255) & ~(size_t)255; };
    const size_t o_iq = al((size_t)a.R * 4), o_sl = o_iq + al((size_t)a.Q * 4), o_fl = o_sl + 2 * PREP_SLOTS * sizeof(unsigned long long);
    if ((s = hg_ensure(c, c->w_cen, o_fl + 256)) != HG_OK) return s;
    auto *cb = static_cast<uint8_t *>(c->w_cen.p);
    auto *info_r = reinterpret_cast<int32_t *>(cb), *info_q = same ? info_r : reinterpret_cast<int32_t *>(cb + o_iq);
    auto *sl_r = reinterpret_cast<unsigned long long *>(cb + o_sl), *sl_q = same ? sl_r : sl_r + PREP_SLOTS;
    auto *fail = reinterpret_cast<uint32_t *>(cb + o_fl), *cverdict = fail + 4;
    HG_HIP(c, hipMemsetAsync(sl_r, 0, 2 * PREP_SLOTS * sizeof(unsigned long long) + 64, c->stream));
    // (zero rows behind the last real one: neither prepass ever writes them, so the raw-value chain below and a repeat
    // call on the same buffer and shape find them still zero)
    if (Rp > a.R && !(c->pad_a_ptr == fa && c->pad_a_rows == a.R && c->pad_a_ldk == ldk)) {
      HG_HIP(c, hipMemsetAsync(fa + (size_t)a.R * ldk, 0, (size_t)(Rp - a.R) * ldk * 2, c->stream));
      c->pad_a_ptr = fa, c->pad_a_rows = a.R, c->pad_a_ldk = ldk;
    }
    if (!same && Qp > a.Q && !(c->pad_b_ptr == fb && c->pad_b_rows == a.Q && c->pad_b_ldk == ldk)) {
      HG_HIP(c, hipMemsetAsync(fb + (size_t)a.Q * ldk, 0, (size_t)(Qp - a.Q) * ldk * 2, c->stream));
      c->pad_b_ptr = fb, c->pad_b_rows = a.Q, c->pad_b_ldk = ldk;
    }
    {
      hg_timed tm(c, HG_T_DIST_PREP);
      hipLaunchKernelGGL(prep_cen_kernel, dim3((a.R + 3) / 4), dim3(256), 0, c->stream, a.ref_hv, a.R, a.hv_d, Kp, ldk, fa, info_r,
                         sl_r, fail, mark);
      HG_HIP(c, hipGetLastError());
      if (!same) {
        hipLaunchKernelGGL(prep_cen_kernel, dim3((a.Q + 3) / 4), dim3(256), 0, c->stream, a.qry_hv, a.Q, a.hv_d, Kp, ldk, fb, info_q,
                           sl_q, fail, mark);
        HG_HIP(c, hipGetLastError());
      }
      hipLaunchKernelGGL(decide_cen_kernel, dim3(1), dim3(256), 0, c->stream, sl_r, sl_q, fail, cverdict, mark);
      HG_HIP(c, hipGetLastError());
    }
    GemmArgs g{};
    g.A = fa, g.B = fb, g.nr = a.ref_n2, g.nq = a.qry_n2, g.R = a.R, g.Q = a.Q, g.Kp = Kp, g.ldk = ldk;
    g.chunk_steps = ~0u, g.kf = (float)a.ksize;
    g.hits = a.hits, g.hit_count = a.hit_count, g.hit_cap = a.hit_cap, g.ani_th = a.ani_th;
    g.symmetric = a.symmetric, g.ref_off = a.ref_off, g.qry_off = a.qry_off;
    g.j_lo = jaccard_lower_bound(a.ani_th, a.ksize);
    if (g.j_lo == -INFINITY) g.pre_c = 0.f, g.pre_b = -INFINITY;
    else if (g.j_lo == INFINITY) g.pre_c = 0.f, g.pre_b = INFINITY;
    else g.pre_c = (float)((double)g.j_lo / (1.0 + (double)g.j_lo) * (1.0 - 1e-5)), g.pre_b = 0.f;
    g.info_r = info_r, g.info_q = info_q, g.hv_d = a.hv_d, g.same_set = same ? 1u : 0u;
    g.verdict = cverdict, g.v_lo = 0, g.v_hi = 0, g.veto = mark;
    int nt = 4;
    {
      const uint64_t tm = (a.R + 255) / 256, ncu = (uint64_t)std::max(c->n_cu, 1);
      const uint64_t r4 = (tm * ((a.Q + 255) / 256) + ncu - 1) / ncu, r5 = (tm * ((a.Q + 319) / 320) + ncu - 1) / ncu;
      if (r5 * 5 < r4 * 4) nt = 5;
      if (c->dbg_dist_tile == "big") nt = 4;
      else if (c->dbg_dist_tile == "wide") nt = 5;
    }
    g.tiles_m = (a.R + 255) / 256, g.tiles_n = (a.Q + (uint32_t)nt * 64 - 1) / ((uint32_t)nt * 64);
    const uint32_t n_tiles = dist_tile_table(c, g, 256, (uint32_t)nt * 64, same && a.ref_off == a.qry_off && c->dbg_dist_order != "plain");
    const size_t lds = nt == 5 ? dist_lds_bytes<true, 5, true>() : dist_lds_bytes<true, 4, true>();
    auto launch_cen = [&](auto kern, const std::string &name, int threads) -> hipError_t {
      const void *fp = reinterpret_cast<const void *>(kern);
      if (std::find(c->lds_attr_done.begin(), c->lds_attr_done.end(), fp) == c->lds_attr_done.end()) {
        const hipError_t e = hipFuncSetAttribute(fp, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        if (e != hipSuccess) return e;
        c->lds_attr_done.push_back(fp);
      }
      c->last_kernel_cen = name;
      hipLaunchKernelGGL(kern, dim3(n_tiles), dim3(threads), lds, c->stream, g);
      return hipGetLastError();
    };
    {
      hg_timed tmg(c, HG_T_DIST, HG_T_DIST_PREP);
      HG_HIP(c, nt == 5 ? launch_cen(HG_DIST_K(false, false, true, true, 5, false, false, false, true), TileCfg<true, 5>::THREADS)
                        : launch_cen(HG_DIST_K(false, false, true, true, 4, false, false, false, true), TileCfg<true, 4>::THREADS));
    }
    veto = mark;
    // the previous call on exactly these operands ran on centred operands: the raw-value chain is not queued again
    if (c->cen_sig_ref == a.ref_hv && c->cen_sig_qry == a.qry_hv && c->cen_sig_r == a.R && c->cen_sig_q == a.Q && c->cen_sig_d == a.hv_d) {
      if (speculated) *speculated = -2;
      return HG_OK;
    }
  }
  auto *st = static_cast<unsigned long long *>(c->w_stats.p);
  // zero rows behind the last real one (tiles hang over); the prepass never writes them, so a repeat call
  // on the same buffer and shape finds them still zero
  if (Rp > a.R && !(c->pad_a_ptr == fa && c->pad_a_rows == a.R && c->pad_a_ldk == ldk)) {
    HG_HIP(c, hipMemsetAsync(fa + (size_t)a.R * ldk, 0, (size_t)(Rp - a.R) * ldk * 2, c->stream));
    c->pad_a_ptr = fa, c->pad_a_rows = a.R, c->pad_a_ldk = ldk;
  }
  if (!same && Qp > a.Q && !(c->pad_b_ptr == fb && c->pad_b_rows == a.Q && c->pad_b_ldk == ldk)) {
    HG_HIP(c, hipMemsetAsync(fb + (size_t)a.Q * ldk, 0, (size_t)(Qp - a.Q) * ldk * 2, c->stream));
    c->pad_b_ptr = fb, c->pad_b_rows = a.Q, c->pad_b_ldk = ldk;
  }
  const size_t plds = (size_t)(2 * (Kp / 64) + 16 + N_CHUNK_CAND) * sizeof(unsigned long long);  // tree + maxima
  // first candidate window that covers all of K (windows are 64 << c dims; beyond the table: none does)
  int c_whole = -1;
  for (int cnd = 0; cnd < N_CHUNK_CAND; ++cnd)
    if ((64u << cnd) >= Kp) {
      c_whole = cnd;
      break;
    }
  unsigned long long h[2 * (1 + N_CHUNK_CAND)];
  const unsigned long long *hr = h, *hq = same ? h : h + 1 + N_CHUNK_CAND;
  int best_c = -1;
  bool fast_done = false, spec = false, spec_win = false;
  (void)hr, (void)hq;
  int spec_cover = -1;  // speculative schedule: highest verdict code with a guarded launch queued
  if (c_whole >= 0) {  // fast prepass: max |x|, the whole-row bound and (win) the 2 048- / 1 024-dim window bounds
    const size_t slot_bytes = PREP_SLOT_VALS * PREP_SLOTS * sizeof(unsigned long long);
    auto *sl = reinterpret_cast<unsigned long long *>(reinterpret_cast<uint8_t *>(st) + 256);
    auto *slq = same ? sl : sl + PREP_SLOT_VALS * PREP_SLOTS;
    const uint32_t win = (Kp % 1024 == 0 && Kp / 1024 <= PREP_MAX_WIN && Kp >= 2048) ? 1u : 0u;
    HG_HIP(c, hipMemsetAsync(sl, 0, 2 * slot_bytes, c->stream));
    {
      hg_timed tm(c, HG_T_DIST_PREP);
      hipLaunchKernelGGL(prep_fast_kernel, dim3((a.R + 3) / 4), dim3(256), 0, c->stream, a.ref_hv, a.R, a.hv_d, Kp, ldk, fa, sl, win, veto);
      HG_HIP(c, hipGetLastError());
      if (!same) {
        hipLaunchKernelGGL(prep_fast_kernel, dim3((a.Q + 3) / 4), dim3(256), 0, c->stream, a.qry_hv, a.Q, a.hv_d, Kp, ldk,
                           fb, slq, win, veto);
        HG_HIP(c, hipGetLastError());
      }
    }
    if (d_verdict && !a.ani_out) {
      // speculative schedule: the verdict is formed on the device and guards the GEMMs queued right behind
      // it; the caller reads it back together with its hit count (no host round trip in between)
      hipLaunchKernelGGL(decide_kernel, dim3(1), dim3(256), 0, c->stream, sl, slq, d_verdict, veto);
      HG_HIP(c, hipGetLastError());
      best_c = c_whole, fast_done = true, spec = true, spec_win = win != 0;
    } else {
      hg_status ps = hg_ensure_pinned(c, 2 * slot_bytes);
      if (ps != HG_OK) return ps;
      auto *hs = static_cast<unsigned long long *>(c->h_pin);
      HG_HIP(c, hipMemcpyAsync(hs, sl, (same ? 1 : 2) * slot_bytes, hipMemcpyDeviceToHost, c->stream));
      HG_HIP(c, hipStreamSynchronize(c->stream));
      unsigned long long mx[2][4] = {{0, 0, 0, 0}, {0, 0, 0, 0}};
      for (int m = 0; m < (same ? 1 : 2); ++m)
        for (uint32_t i = 0; i < PREP_SLOTS; ++i)
          for (uint32_t k = 0; k < PREP_SLOT_VALS; ++k)
            mx[m][k] = std::max(mx[m][k], hs[((size_t)m * PREP_SLOTS + i) * PREP_SLOT_VALS + k]);
      const unsigned long long *r4 = mx[0], *q4 = same ? mx[0] : mx[1];
      auto safe = [](unsigned long long x, unsigned long long y) {
        return x != ~0ull && y != ~0ull && (unsigned __int128)x * y <= ((unsigned __int128)1 << 48);
      };
      if (r4[0] > 2048 || q4[0] > 2048) fast_done = true;  // no f16 path at all: integer kernel below
      else if (safe(r4[1], q4[1])) best_c = c_whole, fast_done = true;
      else if (safe(r4[2], q4[2])) best_c = 5, fast_done = true;  // windows of 2 048 dims (64 << 5)
      else if (safe(r4[3], q4[3])) best_c = 4, fast_done = true;  // windows of 1 024 dims
    }
  }
  if (!fast_done) {  // every candidate window (also rewrites the f16 copies: same values)
    HG_HIP(c, hipMemsetAsync(st, 0, 2 * (1 + N_CHUNK_CAND) * sizeof(unsigned long long), c->stream));
    {
      hg_timed tm(c, HG_T_DIST_PREP);
      hipLaunchKernelGGL(prep_kernel, dim3(a.R), dim3(256), plds, c->stream, a.ref_hv, a.R, a.hv_d, Kp, ldk, fa, st);
      HG_HIP(c, hipGetLastError());
      if (!same) {
        hipLaunchKernelGGL(prep_kernel, dim3(a.Q), dim3(256), plds, c->stream, a.qry_hv, a.Q, a.hv_d, Kp, ldk, fb,
                           st + 1 + N_CHUNK_CAND);
        HG_HIP(c, hipGetLastError());
      }
    }
    HG_HIP(c, hipMemcpyAsync(h, st, sizeof h, hipMemcpyDeviceToHost, c->stream));
    HG_HIP(c, hipStreamSynchronize(c->stream));
    // largest accumulation window whose guaranteed bound sum|r||q| <= sqrt(SR*SQ) stays <= 2^24
    if (hr[0] <= 2048 && hq[0] <= 2048) {
      for (int cnd = N_CHUNK_CAND - 1; cnd >= 0; --cnd) {
        const unsigned __int128 prod = (unsigned __int128)hr[1 + cnd] * hq[1 + cnd];
        if (prod <= ((unsigned __int128)1 << 48)) {
          best_c = cnd;
          break;
        }
      }
    }
  }
  const float kf = (float)a.ksize;
  hg_timed tm(c, HG_T_DIST);
  c->last_dist_path = best_c < 0 ? 2 : 0;  // (a valid i8 attempt overrides this after the caller's read-back)
  if (best_c < 0) {  // values too large for the f16 path: exact integer kernel
    dim3 grid((a.Q + FB_T - 1) / FB_T, (a.R + FB_T - 1) / FB_T);
    c->last_kernel[HG_T_DIST] = "dist_int_kernel";
    hipLaunchKernelGGL(dist_int_kernel, grid, dim3(FB_T * FB_T), 0, c->stream, a.ref_hv, a.qry_hv, a, kf);
    HG_HIP(c, hipGetLastError());
    return HG_OK;
  }
  // one GEMM launch for accumulation windows of 64 << bc dims; guard != nullptr: runs only if
  // v_lo <= guard[0] <= v_hi, and (from_verdict) takes its window length from guard[1]
  auto gemm = [&](int bc, const uint32_t *guard, uint32_t v_lo, uint32_t v_hi, bool from_verdict) -> hg_status {
  const int best_c = bc;
  GemmArgs g{};
  g.A = fa, g.B = fb, g.nr = a.ref_n2, g.nq = a.qry_n2;
  g.R = a.R, g.Q = a.Q, g.Kp = Kp, g.ldk = ldk;
  g.chunk_steps = (64u << best_c) / BK;
  g.kf = kf;
  g.ani_out = a.ani_out, g.hits = a.hits, g.hit_count = a.hit_count, g.hit_cap = a.hit_cap;
  g.ani_th = a.ani_th, g.symmetric = a.symmetric, g.ref_off = a.ref_off, g.qry_off = a.qry_off;
  g.verdict = guard, g.v_lo = v_lo, g.v_hi = v_hi, g.chunk_from_verdict = from_verdict ? 1u : 0u;
  g.veto = veto;
  g.j_lo = jaccard_lower_bound(a.ani_th, a.ksize);
  if (g.j_lo == -INFINITY) g.pre_c = 0.f, g.pre_b = -INFINITY;       // everything goes on to phase 1
  else if (g.j_lo == INFINITY) g.pre_c = 0.f, g.pre_b = INFINITY;    // ANI <= 100 < ani_th: nothing does
  else g.pre_c = (float)((double)g.j_lo / (1.0 + (double)g.j_lo) * (1.0 - 1e-5)), g.pre_b = 0.f;
  const bool whole_k = (64u << best_c) >= Kp;  // one window covers K: no i32 side accumulators
  const bool full = a.ani_out != nullptr;
  // big tiles when the problem fills the chip with them (Rp, Qp are multiples of 128: the last big
  // tile may hang over by 128 rows, which the zero padding of the operand copies must cover)
  bool big = !full && whole_k && (uint64_t)a.R * a.Q >= (uint64_t)256 * 256 * 256;
  int nt = 4;  // 16-column MFMA tiles per wave: 4 -> 256-wide tiles, 5 -> 320-wide
  if (big) {   // the width that needs fewer rounds over the CUs (a round of 320-wide tiles costs 5/4)
    const uint64_t tm = (a.R + 255) / 256, ncu = (uint64_t)std::max(c->n_cu, 1);
    const uint64_t r4 = (tm * ((a.Q + 255) / 256) + ncu - 1) / ncu, r5 = (tm * ((a.Q + 319) / 320) + ncu - 1) / ncu;
    if (r5 * 5 < r4 * 4) nt = 5;
  }
  if (const char *e = c->dbg_dist_tile.empty() ? nullptr : c->dbg_dist_tile.c_str()) {  // test hook (hg_ctx_set_debug): force a geometry
    if (!std::strcmp(e, "big")) big = !full && whole_k, nt = 4;
    else if (!std::strcmp(e, "wide")) big = !full && whole_k, nt = 5;
    else if (!std::strcmp(e, "small")) big = false;
  }
  if (!big) nt = 4;
  // several exact f32 windows per row (sketches of more than ~4 000 hashes at D = 4096): the i32 side
  // accumulators double the accumulator registers, so the 256-row geometry narrows to 64 * NT_CHUNKED columns
  constexpr int NT_CHUNKED = 3;
  bool big_chunked = !full && !whole_k && (uint64_t)a.R * a.Q >= (uint64_t)256 * 256 * 256;
  if (const char *e = c->dbg_dist_tile.empty() ? nullptr : c->dbg_dist_tile.c_str()) {
    if (!std::strcmp(e, "small")) big_chunked = false;
    else if (!std::strcmp(e, "big") || !std::strcmp(e, "wide")) big_chunked = !full && !whole_k;
  }
  if (big_chunked) big = true, nt = NT_CHUNKED;
  const uint32_t bm = big ? 256 : 128, bn = big ? (uint32_t)nt * 64 : 128;
  g.tiles_m = (a.R + bm - 1) / bm, g.tiles_n = (a.Q + bn - 1) / bn;
  // (thresholded self-comparison: the tiles on the diagonal first, as on the i8 path)
  const uint32_t n_tiles = dist_tile_table(c, g, bm, bn, !full && same && a.ref_off == a.qry_off && c->dbg_dist_order != "plain");
  auto launch = [&](auto kern, const std::string &name, int threads, size_t lds) -> hipError_t {
    if (!guard || v_lo == 0) c->last_kernel[HG_T_DIST] = name;  // (a guarded second launch covers verdicts 1..2 only)
    const void *fp = reinterpret_cast<const void *>(kern);
    if (std::find(c->lds_attr_done.begin(), c->lds_attr_done.end(), fp) == c->lds_attr_done.end()) {
      hipError_t e = hipFuncSetAttribute(fp, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
      if (e != hipSuccess) return e;
      c->lds_attr_done.push_back(fp);
    }
    hipLaunchKernelGGL(kern, dim3(n_tiles), dim3(threads), lds, c->stream, g);
    return hipGetLastError();
  };
  hipError_t le;
  // (two operand stages or the epilogue's candidate lists, + the tile's row / column words: dist_lds_bytes)
  const size_t lds_small = dist_lds_bytes<false, 4, false>(), lds_dma = dist_lds_bytes<true, 4, true>();
  const size_t lds_wide = dist_lds_bytes<true, 5, true>(), lds_chunked = dist_lds_bytes<true, NT_CHUNKED, true>();
  static_assert(dist_lds_bytes<true, 5, true>() <= 160 * 1024, "the widest tile fits the CU's LDS");
  if (big_chunked) le = launch(HG_DIST_K(true, false, true, true, NT_CHUNKED), TileCfg<true, NT_CHUNKED>::THREADS, lds_chunked);
  else if (big && nt == 5) le = launch(HG_DIST_K(false, false, true, true, 5), TileCfg<true, 5>::THREADS, lds_wide);
  else if (big) le = launch(HG_DIST_K(false, false, true, true), TileCfg<true>::THREADS, lds_dma);
  else if (whole_k && full) le = launch(HG_DIST_K(false, true, false), TileCfg<false>::THREADS, lds_small);
  else if (whole_k) le = launch(HG_DIST_K(false, false, false), TileCfg<false>::THREADS, lds_small);
  else if (full) le = launch(HG_DIST_K(true, true, false), TileCfg<false>::THREADS, lds_small);
  else le = launch(HG_DIST_K(true, false, false), TileCfg<false>::THREADS, lds_small);
  HG_HIP(c, le);
  return HG_OK;
  };
  if (!spec) return gemm(best_c, nullptr, 0, 0, false);
  // speculative: the one-window kernel for verdict 0 and, where the prepass measured the 2 048 / 1 024 windows,
  // the windowed kernel for verdicts 1..2 right behind it (whichever is not chosen returns at once)
  hg_status gs = gemm(c_whole, d_verdict, 0, 0, false);
  if (gs != HG_OK) return gs;
  spec_cover = 0;
  if (spec_win && Kp > 2048) {
    if ((gs = gemm(4, d_verdict, 1, 2, true)) != HG_OK) return gs;
    spec_cover = 2;
  }
  if (speculated) *speculated = spec_cover;
  return HG_OK;
}

// ---- operands prepared where the rows live (sharded callers) ---------------------------------------------------------
size_t hg_dist_ops_row_bytes_impl(uint32_t hv_d) { return (size_t)((hv_d + 127) / 128 * 128) + 128; }
size_t hg_dist_ops_meta_bytes_impl() { return sizeof(I8RowMeta); }
size_t hg_dist_ops_padded_rows_impl(size_t n) {
  return std::max(std::max((n + 255) / 256 * 256, (n + 319) / 320 * 320), (n + 191) / 192 * 192);
}
hg_status hg_run_dist_prep_ops(hg_ctx *c, const int16_t *d_hv, uint32_t rows, uint32_t hv_d, uint8_t *d_ops, void *d_meta,
                               uint32_t *d_flag) {
  if (hv_d > 8192 || hv_d % 8) return hg_fail(c, HG_ERR_UNSUPPORTED, "prepared operands need hv_d <= 8192, hv_d % 8 == 0");
  const uint32_t kp8 = (hv_d + 127) / 128 * 128, ldk8 = kp8 + 128;
  HG_HIP(c, hipMemsetAsync(d_flag, 0, sizeof(uint32_t), c->stream));
  hg_timed tmp(c, HG_T_DIST_PREP);
  // (ctrl[1] is where the kernel ORs its failure bits: the caller's flag word)
  hipLaunchKernelGGL(prep_i8_kernel, dim3((rows + 3) / 4), dim3(256), 0, c->stream, d_hv, rows, hv_d, kp8, ldk8,
                     reinterpret_cast<int8_t *>(d_ops), static_cast<int32_t *>(nullptr), static_cast<int32_t *>(nullptr),
                     static_cast<uint32_t *>(nullptr), static_cast<I8Outlier *>(nullptr), 0u, d_flag - 1, 0u,
                     static_cast<I8RowMeta *>(d_meta));
  HG_HIP(c, hipGetLastError());
  return HG_OK;
}

#ifdef HG_DIST_STAMPS
extern "C" int hg_debug_dist_tile_real(unsigned long long *out /* 2048 * 2 */) {
  return (int)hipMemcpyFromSymbol(out, HIP_SYMBOL(g_dist_tile_real), sizeof(unsigned long long) * 2048 * 2);
}
extern "C" int hg_debug_dist_tile_all(unsigned long long *out /* 2048 * 16 */) {
  return (int)hipMemcpyFromSymbol(out, HIP_SYMBOL(g_dist_tile_all), sizeof(unsigned long long) * 2048 * 16);
}
extern "C" int hg_debug_dist_tile_stamps(unsigned long long *out /* 16 * 8 * 10 */) {
  return (int)hipMemcpyFromSymbol(out, HIP_SYMBOL(g_dist_tile_stamps), sizeof(unsigned long long) * 16 * 8 * 10);
}
extern "C" int hg_debug_dist_stamps(unsigned long long *out /* 16 * 2 * 8 * 6 */) {
  return (int)hipMemcpyFromSymbol(out, HIP_SYMBOL(g_dist_stamps), sizeof(unsigned long long) * 16 * 2 * 8 * 6);
}
#endif

// ---- the launch order, for inspection (host only) ----------------------------------------------------------------------
extern "C" hg_status hg_dist_tile_order(uint32_t tiles_m, uint32_t tiles_n, uint32_t tile_rows, uint32_t tile_cols, int diagonal_first,
                                        int symmetric, uint64_t ref_off, uint64_t qry_off, uint32_t *out, size_t cap, size_t *n_slots) {
  if (!n_slots || !tile_rows || !tile_cols || tiles_m > 0xFFFFu || tiles_n > 0xFFFEu || (uint64_t)tiles_m * tiles_n > (1u << 24)) return HG_ERR_INVALID;
  try {
    const std::vector<uint32_t> t = build_tile_order(tiles_m, tiles_n, tile_rows, tile_cols, diagonal_first != 0, symmetric != 0, ref_off, qry_off);
    *n_slots = t.size();
    if (t.size() > cap || (!out && !t.empty())) return HG_ERR_CAPACITY;
    std::memcpy(out, t.data(), t.size() * sizeof(uint32_t));
  } catch (const std::bad_alloc &) {
    return HG_ERR_OOM;
  }
  return HG_OK;
}

// ---- the ANI formula and its logarithm on their own (diagnostics; tests/test_gpu_ani_exact.py) ------------------------
namespace {
__global__ void logf_kernel(const float *__restrict__ x, uint32_t first_bits, size_t n, float *__restrict__ out) {
  const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) out[i] = hg_logf(x ? x[i] : __uint_as_float(first_bits + (uint32_t)i));
}
__global__ void ani_from_dots_kernel(const int32_t *__restrict__ dot, const int32_t *__restrict__ nr, const int32_t *__restrict__ nq,
                                     size_t n, float kf, float *__restrict__ out) {
  const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) out[i] = ani_from_dot(dot[i], nr[i], nq[i], kf);
}
}  // namespace

extern "C" hg_status hg_logf_dev(hg_ctx *c, const float *d_x, uint32_t first_bits, size_t n, float *d_out) {
  if (!c) return HG_ERR_INVALID;
  if (n == 0) return HG_OK;
  if (!d_out || n > ((size_t)1 << 32)) return hg_fail(c, HG_ERR_INVALID, "hg_logf_dev: bad argument");
  HG_ENTER(c);
  hipLaunchKernelGGL(logf_kernel, dim3((uint32_t)((n + 255) / 256)), dim3(256), 0, c->stream, d_x, first_bits, n, d_out);
  HG_HIP(c, hipGetLastError());
  return HG_OK;
}

extern "C" hg_status hg_ani_from_dots_dev(hg_ctx *c, const int32_t *d_dot, const int32_t *d_norm2_r, const int32_t *d_norm2_q, size_t n,
                                          uint32_t ksize, float *d_ani) {
  if (!c) return HG_ERR_INVALID;
  if (n == 0) return HG_OK;
  if (!d_dot || !d_norm2_r || !d_norm2_q || !d_ani || ksize == 0 || n > ((size_t)1 << 32)) return hg_fail(c, HG_ERR_INVALID, "hg_ani_from_dots_dev: bad argument");
  HG_ENTER(c);
  hipLaunchKernelGGL(ani_from_dots_kernel, dim3((uint32_t)((n + 255) / 256)), dim3(256), 0, c->stream, d_dot, d_norm2_r, d_norm2_q, n,
                     (float)ksize, d_ani);
  HG_HIP(c, hipGetLastError());
  return HG_OK;
}
