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
// glibc by <= 1 ulp.
#include <algorithm>
#include <cmath>
#include <cstdlib>
#include <cstring>
#include <type_traits>
#include <utility>
#include <vector>

#include "hg_internal.h"

namespace {

typedef _Float16 half8 __attribute__((ext_vector_type(8)));
typedef float float4v __attribute__((ext_vector_type(4)));

// ---- ANI epilogue (src/dist.rs:153-160) -----------------------------------------------------
__device__ __forceinline__ float ani_from_dot(int32_t dot, int32_t nr, int32_t nq, float kf) {
  const int32_t den = (int32_t)((uint32_t)nr + (uint32_t)nq - (uint32_t)dot);  // i32 wrapping
  const float jaccard = (float)dot / (float)den;
  const float inner = 1.0f / jaccard + 1.0f;
  const float x = 2.0f / inner;
  float ani = 1.0f + logf(x) / kf;
  if (ani != ani) return 0.0f;  // is_nan -> 0
  ani = fminf(ani, 1.0f);
  ani = fmaxf(ani, 0.0f);
  return ani * 100.0f;
}

// ---- prepass: i16 -> f16 (zero padded to Kp) + exactness statistics --------------------------
// stats[0]            = max |x|
// stats[1 + c]        = max over rows and aligned chunks of 64<<c dims of sum x^2   (c = 0..7)
constexpr int N_CHUNK_CAND = 8;  // 64 .. 8192
// One workgroup per row; each lane converts 8 consecutive values per trip (16-byte loads and stores), so
// a 64-dim block is 8 adjacent lanes.  Everything per element is packed 16-bit or dot2 work straight on
// the loaded words: |x| by v_pk_sub/v_pk_max, sum x^2 by v_dot2_i32_i16, the 8-lane block sum by three
// DPP adds.  Block sums are 32-bit: exact whenever |x| <= 2048 (64 * 2^22 = 2^28), and when some |x| is
// larger the f16 path is abandoned anyway (stats[0] decides first).  The chunk maxima for all candidate
// window sizes come from a pairwise-sum tree over the block sums in LDS with one LDS atomic max per
// level -- no cross-lane shuffles (the first version spent most of its time in ~100 dependent
// ds_bpermute reductions per row: 0.115 ms for 10 000 rows against 0.04 ms of memory time).
typedef short short2v __attribute__((ext_vector_type(2)));
template <int CTRL>
__device__ __forceinline__ int dpp_add(int v) {  // v + v[lane permuted by CTRL]
  return v + __builtin_amdgcn_update_dpp(0, v, CTRL, 0xf, 0xf, true);
}
__global__ __launch_bounds__(256) void prep_kernel(const int16_t *__restrict__ hv, uint32_t rows,
                                                   uint32_t hv_d, uint32_t kp, uint32_t ldk,
                                                   _Float16 *__restrict__ out,
                                                   unsigned long long *__restrict__ stats) {
  extern __shared__ unsigned long long s_lv[];  // tree levels: nblk, ceil(nblk/2), ... 1 values, then N_CHUNK_CAND maxima
  __shared__ uint32_t s_max;
  const uint32_t row = blockIdx.x;
  const int16_t *__restrict__ src = hv + (size_t)row * hv_d;
  _Float16 *__restrict__ dst = out + (size_t)row * ldk;
  const uint32_t nblk = kp / 64;
  uint32_t tree = 0;  // total tree size
  for (uint32_t n = nblk;; n = (n + 1) / 2) {
    tree += n;
    if (n == 1) break;
  }
  unsigned long long *s_lvmax = s_lv + tree;
  if (threadIdx.x == 0) s_max = 0;
  if (threadIdx.x < N_CHUNK_CAND) s_lvmax[threadIdx.x] = 0;
  __syncthreads();
  uint32_t mxpk = 0;  // packed running max of |x| (two u16 lanes)
  const bool vec_ok = (hv_d % 8 == 0) && ((reinterpret_cast<uintptr_t>(src) & 15) == 0);
  for (uint32_t d0 = threadIdx.x * 8; d0 < kp; d0 += blockDim.x * 8) {
    uint32_t w[4];
    if (vec_ok && d0 + 8 <= hv_d) {
      const uint4 raw = *reinterpret_cast<const uint4 *>(src + d0);
      w[0] = raw.x, w[1] = raw.y, w[2] = raw.z, w[3] = raw.w;
    } else {
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        const uint32_t lo = (d0 + 2 * i < hv_d) ? (uint16_t)src[d0 + 2 * i] : 0u;
        const uint32_t hi = (d0 + 2 * i + 1 < hv_d) ? (uint16_t)src[d0 + 2 * i + 1] : 0u;
        w[i] = lo | (hi << 16);
      }
    }
    half8 h;
    int sq = 0;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      short2v x2;
      __builtin_memcpy(&x2, &w[i], 4);
      sq = __builtin_amdgcn_sdot2(x2, x2, sq, false);
      const short2v ab = __builtin_elementwise_max(x2, (short2v)(-x2));  // |x| (-32768 stays 0x8000: larger than any u16 <= 2048)
      uint32_t abw;
      __builtin_memcpy(&abw, &ab, 4);
      typedef unsigned short ushort2v __attribute__((ext_vector_type(2)));
      ushort2v m0, m1;
      __builtin_memcpy(&m0, &mxpk, 4);
      __builtin_memcpy(&m1, &abw, 4);
      m0 = __builtin_elementwise_max(m0, m1);
      __builtin_memcpy(&mxpk, &m0, 4);
      h[2 * i] = (_Float16)x2.x;
      h[2 * i + 1] = (_Float16)x2.y;
    }
    *reinterpret_cast<half8 *>(dst + d0) = h;
    sq = dpp_add<0xB1>(sq);   // quad_perm [1,0,3,2]
    sq = dpp_add<0x4E>(sq);   // quad_perm [2,3,0,1]
    sq = dpp_add<0x141>(sq);  // row_half_mirror: the other quad of the 8-lane group
    if ((threadIdx.x & 7) == 0) s_lv[d0 / 64] = (unsigned long long)(uint32_t)sq;
  }
  const uint32_t mx = (mxpk & 0xffffu) > (mxpk >> 16) ? (mxpk & 0xffffu) : (mxpk >> 16);
  if (mx) atomicMax(&s_max, mx);
  __syncthreads();
  // same-address device atomics serialise at ~12 ns each: only the few rows that raise a maximum
  // issue one (a relaxed agent-scope load may be stale, which at worst costs a redundant atomic)
  auto raise = [](unsigned long long *p, unsigned long long v) {
    if (v > __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) atomicMax(p, v);
  };
  if (threadIdx.x == 0) raise(&stats[0], (unsigned long long)s_max);
  // level c holds the sums of aligned chunks of 2^c blocks (the last one may be partial)
  unsigned long long *lv = s_lv;
  uint32_t n = nblk;
  for (int c = 0; c < N_CHUNK_CAND; ++c) {
    unsigned long long best = 0;
    for (uint32_t i = threadIdx.x; i < n; i += blockDim.x) best = lv[i] > best ? lv[i] : best;
    if (best) atomicMax(&s_lvmax[c], best);
    if (n > 1) {  // build the next level
      unsigned long long *nx = lv + n;
      const uint32_t n2 = (n + 1) / 2;
      for (uint32_t i = threadIdx.x; i < n2; i += blockDim.x) nx[i] = lv[2 * i] + (2 * i + 1 < n ? lv[2 * i + 1] : 0ull);
      lv = nx, n = n2;
    }
    __syncthreads();
  }
  if (threadIdx.x < N_CHUNK_CAND && s_lvmax[threadIdx.x]) raise(&stats[1 + threadIdx.x], s_lvmax[threadIdx.x]);
}

// Fast prepass for the common case: conversion plus only max |x| and the maximum whole-row sum of squares
// (the statistic that decides whether ONE f32 accumulation window covers K).  One wave per row, four rows
// per workgroup, no LDS and no barrier: all of a row's 16-byte loads are in flight together, the two row
// statistics are reduced with DPP.  If the whole-row bound turns out unsafe, hg_run_dist runs prep_kernel
// (all candidate windows) as a second pass.
constexpr uint32_t PREP_SLOTS = 1024, PREP_SLOT_VALS = 4, PREP_MAX_WIN = 32;
template <int CTRL, int ROW_MASK = 0xf>
__device__ __forceinline__ uint32_t dpp_get(uint32_t v) {  // v[lane permuted by CTRL], 0 where nothing arrives
  return (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, CTRL, ROW_MASK, 0xf, true);
}
template <int CTRL, int ROW_MASK = 0xf>
__device__ __forceinline__ unsigned long long dpp_get64(unsigned long long v) {
  return (unsigned long long)dpp_get<CTRL, ROW_MASK>((uint32_t)v) |
         ((unsigned long long)dpp_get<CTRL, ROW_MASK>((uint32_t)(v >> 32)) << 32);
}
__global__ __launch_bounds__(256) void prep_fast_kernel(const int16_t *__restrict__ hv, uint32_t rows,
                                                        uint32_t hv_d, uint32_t kp, uint32_t ldk,
                                                        _Float16 *__restrict__ out,
                                                        unsigned long long *__restrict__ slots, uint32_t win,
                                                        const uint32_t *__restrict__ veto) {
  if (veto && veto[0] != 0u) return;  // a path queued before this prepass (i8 or centred f16 operands) did the work
  // win != 0 (kp a multiple of 1024, at most PREP_MAX_WIN windows): the row's sum of squares per aligned
  // 1024-dim window is collected too (per-wave LDS accumulators), for the 2 048- and 1 024-dim bounds
  __shared__ unsigned long long s_win[4][PREP_MAX_WIN];
  const uint32_t lane = threadIdx.x & 63, wv = threadIdx.x >> 6, row = blockIdx.x * 4 + wv;
  if (row >= rows) return;  // whole wave
  volatile unsigned long long *mywin = s_win[wv];
  if (win && lane < PREP_MAX_WIN) mywin[lane] = 0;
  const int16_t *__restrict__ src = hv + (size_t)row * hv_d;
  _Float16 *__restrict__ dst = out + (size_t)row * ldk;
  const bool vec_ok = (hv_d % 8 == 0) && ((reinterpret_cast<uintptr_t>(src) & 15) == 0);
  uint32_t mxpk = 0, sq = 0;  // per lane <= 512 squares <= 2^22 each when |x| <= 2048
  auto fetch = [&](uint32_t d0, uint32_t w[4]) {
    if (d0 >= kp) {
      w[0] = w[1] = w[2] = w[3] = 0u;
    } else if (vec_ok && d0 + 8 <= hv_d) {
      const uint4 raw = *reinterpret_cast<const uint4 *>(src + d0);
      w[0] = raw.x, w[1] = raw.y, w[2] = raw.z, w[3] = raw.w;
    } else {
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        const uint32_t lo = (d0 + 2 * i < hv_d) ? (uint16_t)src[d0 + 2 * i] : 0u;
        const uint32_t hi = (d0 + 2 * i + 1 < hv_d) ? (uint16_t)src[d0 + 2 * i + 1] : 0u;
        w[i] = lo | (hi << 16);
      }
    }
  };
  // Four 16-byte loads in flight per lane.  The 512-dim chunks of a row are visited in an order rotated by
  // the row index: with the natural order every resident wave would be at the same column offset of its
  // row at the same time, and with a power-of-two row pitch (8 KiB at D = 4096) those addresses all fall
  // on the same few memory channels (measured: 94 us instead of 30 us for 10 000 rows).
  const uint32_t nchunks = (kp + 511) / 512;
  for (uint32_t q = 0; q < nchunks; q += 4) {
    uint32_t w[4][4], d0[4];
#pragma unroll
    for (int t = 0; t < 4; ++t) {
      d0[t] = q + t < nchunks ? ((q + t + row) % nchunks) * 512 + lane * 8 : kp;
      fetch(d0[t], w[t]);
    }
#pragma unroll
    for (int t = 0; t < 4; ++t) {
      half8 h;
      uint32_t sqc = 0;  // this lane's share of the chunk
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        short2v x2;
        __builtin_memcpy(&x2, &w[t][i], 4);
        sqc = (uint32_t)__builtin_amdgcn_sdot2(x2, x2, (int)sqc, false);
        const short2v ab = __builtin_elementwise_max(x2, (short2v)(-x2));
        typedef unsigned short ushort2v __attribute__((ext_vector_type(2)));
        ushort2v m0, m1;
        __builtin_memcpy(&m0, &mxpk, 4);
        __builtin_memcpy(&m1, &ab, 4);
        m0 = __builtin_elementwise_max(m0, m1);
        __builtin_memcpy(&mxpk, &m0, 4);
        h[2 * i] = (_Float16)x2.x;
        h[2 * i + 1] = (_Float16)x2.y;
      }
      if (d0[t] < kp) *reinterpret_cast<half8 *>(dst + d0[t]) = h;
      sq += sqc;
      if (win) {  // row-of-16 sums by DPP, then four LDS adds per chunk instead of 64 on one address
        uint32_t rs = sqc;
        rs += dpp_get<0xB1>(rs), rs += dpp_get<0x4E>(rs), rs += dpp_get<0x141>(rs), rs += dpp_get<0x140>(rs);
        if ((lane & 15) == 0 && d0[t] < kp)
          atomicAdd(const_cast<unsigned long long *>(&mywin[d0[t] >> 10]), (unsigned long long)rs);
      }
    }
  }
  uint32_t mx = (mxpk & 0xffffu) > (mxpk >> 16) ? (mxpk & 0xffffu) : (mxpk >> 16);
  unsigned long long sum = sq;
  // butterfly inside each row of 16 lanes, then row 0 -> 1, 2 -> 3 (row_bcast15), rows 0..1 -> 2..3 (row_bcast31)
#define HG_STEP(CTRL)                                  \
  {                                                    \
    const uint32_t om = dpp_get<CTRL>(mx);             \
    mx = om > mx ? om : mx;                            \
    sum += dpp_get64<CTRL>(sum);                       \
  }
  HG_STEP(0xB1) HG_STEP(0x4E) HG_STEP(0x141) HG_STEP(0x140)
#undef HG_STEP
  {
    const uint32_t om = dpp_get<0x142, 0xa>(mx);
    mx = om > mx ? om : mx;
    sum += dpp_get64<0x142, 0xa>(sum);
  }
  {
    const uint32_t om = dpp_get<0x143, 0xc>(mx);
    mx = om > mx ? om : mx;
    sum += dpp_get64<0x143, 0xc>(sum);
  }
  // window maxima of this row: lane w holds window w (LDS is in order per wave: the adds above are done)
  unsigned long long w1 = (win && lane < kp / 1024) ? mywin[lane] : 0ull;
  unsigned long long w2 = w1 + dpp_get64<0xB1>(w1);  // aligned pairs of 1 024-windows = 2 048-windows
#define HG_MAXSTEP(CTRL, MASK)                                   \
  {                                                              \
    const unsigned long long o1 = dpp_get64<CTRL, MASK>(w1), o2 = dpp_get64<CTRL, MASK>(w2); \
    w1 = o1 > w1 ? o1 : w1, w2 = o2 > w2 ? o2 : w2;              \
  }
  HG_MAXSTEP(0xB1, 0xf) HG_MAXSTEP(0x4E, 0xf) HG_MAXSTEP(0x141, 0xf) HG_MAXSTEP(0x140, 0xf) HG_MAXSTEP(0x142, 0xa) HG_MAXSTEP(0x143, 0xc)
#undef HG_MAXSTEP
  if (lane == 63) {
    auto raise = [](unsigned long long *p, unsigned long long v) {
      if (v > __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) atomicMax(p, v);
    };
    // {max |x|, max row sum, max 2 048-window sum, max 1 024-window sum} per slot; the maximum over the slots is
    // taken afterwards.  (One shared set of counters cost ~55 us per launch: the ~8 000 waves resident at the
    // start all see the initial zero and all issue their atomics to the same address, ~12 ns each.)
    unsigned long long *sl = slots + PREP_SLOT_VALS * (blockIdx.x % PREP_SLOTS);
    raise(&sl[0], (unsigned long long)mx);
    raise(&sl[1], sum);
    raise(&sl[2], win ? w2 : ~0ull);
    raise(&sl[3], win ? w1 : ~0ull);
  }
}

// Exactness verdict of the fast prepass, on the device.  verdict[0]: 0 = |x| <= 2048 everywhere and, by
// Cauchy-Schwarz, every dot product is exact in ONE f32 accumulation window; 1 / 2 = exact with windows of
// 2 048 / 1 024 dims (verdict[1] = window length in K-steps of 64); 3 = none of these.
__device__ __forceinline__ bool window_safe(unsigned long long a, unsigned long long b) {
  return a != ~0ull && b != ~0ull && (unsigned __int128)a * b <= ((unsigned __int128)1 << 48);
}
__global__ __launch_bounds__(256) void decide_kernel(const unsigned long long *__restrict__ slots_r,
                                                     const unsigned long long *__restrict__ slots_q,
                                                     uint32_t *__restrict__ verdict, const uint32_t *__restrict__ veto) {
  __shared__ unsigned long long s_red[8][256];
  if (veto && veto[0] != 0u) {  // uniform: the i8 / centred f16 path did the work; report "covered" to the host
    if (threadIdx.x == 0) verdict[0] = 0, verdict[1] = 0;
    return;
  }
  unsigned long long v[8] = {0, 0, 0, 0, 0, 0, 0, 0};
  for (uint32_t i = threadIdx.x; i < PREP_SLOTS; i += 256) {
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      v[k] = max(v[k], slots_r[PREP_SLOT_VALS * i + k]);
      v[4 + k] = max(v[4 + k], slots_q[PREP_SLOT_VALS * i + k]);
    }
  }
#pragma unroll
  for (int k = 0; k < 8; ++k) s_red[k][threadIdx.x] = v[k];
  __syncthreads();
  for (uint32_t o = 128; o > 0; o >>= 1) {
    if (threadIdx.x < o) {
#pragma unroll
      for (int k = 0; k < 8; ++k) s_red[k][threadIdx.x] = max(s_red[k][threadIdx.x], s_red[k][threadIdx.x + o]);
    }
    __syncthreads();
  }
  if (threadIdx.x == 0) {
    uint32_t code = 3, steps = 0;
    if (s_red[0][0] <= 2048 && s_red[4][0] <= 2048) {
      if (window_safe(s_red[1][0], s_red[5][0])) code = 0;
      else if (window_safe(s_red[2][0], s_red[6][0])) code = 1, steps = 2048 / 64;
      else if (window_safe(s_red[3][0], s_red[7][0])) code = 2, steps = 1024 / 64;
    }
    verdict[0] = code, verdict[1] = steps;
  }
}

// ---- centred f16 operands ------------------------------------------------------------------------------------------
// One wave per row (like prep_fast_kernel): c = (x + e) >> 1 with e = the row's parity, written as f16 (exact for
// |c| <= 2048), the row's info word 2 S + e, and per slot the maximum row sum of c^2 -- the statistic that proves ONE f32
// accumulation window exact (sum |c_r||c_q| <= sqrt(sum c_r^2 sum c_q^2) <= 2^24).  A row of mixed parity, or |c| > 2048,
// raises `fail`.  skip: words that switch the kernel off when a path queued in front already did the work.
__global__ __launch_bounds__(256) void prep_cen_kernel(const int16_t *__restrict__ hv, uint32_t rows, uint32_t hv_d, uint32_t kp,
                                                       uint32_t ldk, _Float16 *__restrict__ out, int32_t *__restrict__ rowinfo,
                                                       unsigned long long *__restrict__ slots, uint32_t *__restrict__ fail,
                                                       const uint32_t *__restrict__ skip) {
  if (skip && skip[0] != 0u) return;
  const uint32_t lane = threadIdx.x & 63, wv = threadIdx.x >> 6, row = blockIdx.x * 4 + wv;
  if (row >= rows) return;  // whole wave
  const int16_t *__restrict__ src = hv + (size_t)row * hv_d;
  _Float16 *__restrict__ dst = out + (size_t)row * ldk;
  const int32_t e = (int32_t)src[0] & 1;
  const bool vec_ok = (hv_d % 8 == 0) && ((reinterpret_cast<uintptr_t>(src) & 15) == 0);
  int32_t S = 0;
  unsigned long long sq = 0;
  uint32_t bad = 0;
  const uint32_t nchunks = (kp + 511) / 512;
  for (uint32_t q = 0; q < nchunks; ++q) {
    const uint32_t d0 = ((q + row) % nchunks) * 512 + lane * 8;  // (chunk order rotated by the row: see prep_fast_kernel)
    if (d0 >= kp) continue;
    int32_t x[8];
    if (vec_ok && d0 + 8 <= hv_d) {
      const uint4 raw = *reinterpret_cast<const uint4 *>(src + d0);
      const uint32_t w[4] = {raw.x, raw.y, raw.z, raw.w};
#pragma unroll
      for (int i = 0; i < 4; ++i) x[2 * i] = (int16_t)(w[i] & 0xffffu), x[2 * i + 1] = (int16_t)(w[i] >> 16);
    } else {
#pragma unroll
      for (int i = 0; i < 8; ++i) x[i] = d0 + i < hv_d ? (int32_t)src[d0 + i] : -e;  // padding: c = 0
    }
    half8 h;
#pragma unroll
    for (int i = 0; i < 8; ++i) {
      if (d0 + i < hv_d && ((x[i] ^ e) & 1)) bad |= 1u;  // mixed parity
      const int32_t c = (x[i] + e) >> 1;
      if (c > 2048 || c < -2048) bad |= 2u;
      S += c;
      sq += (unsigned long long)((long long)c * c);
      h[i] = (_Float16)c;
    }
    *reinterpret_cast<half8 *>(dst + d0) = h;
  }
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) S += __shfl_xor(S, o), sq += __shfl_xor(sq, o);
  const bool anybad = __any(bad != 0);
  if (lane == 0) {
    rowinfo[row] = 2 * S + e;
    unsigned long long *sl = slots + (blockIdx.x % PREP_SLOTS);
    if (sq > __hip_atomic_load(sl, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) atomicMax(sl, sq);
    if (anybad) atomicOr(fail, 1u);
  }
}
// verdict of the centred path: mark[0] <- 2 ("the centred f16 kernel does the work": the kernels queued behind it return)
// and verdict[0] <- 0 iff no path in front did the work, no row failed and one window is exact; else verdict[0] <- 3
__global__ __launch_bounds__(256) void decide_cen_kernel(const unsigned long long *__restrict__ slots_r,
                                                         const unsigned long long *__restrict__ slots_q,
                                                         const uint32_t *__restrict__ fail, uint32_t *__restrict__ verdict,
                                                         uint32_t *__restrict__ mark) {
  __shared__ unsigned long long s_red[2][256];
  unsigned long long a = 0, b = 0;
  for (uint32_t i = threadIdx.x; i < PREP_SLOTS; i += 256) a = max(a, slots_r[i]), b = max(b, slots_q[i]);
  s_red[0][threadIdx.x] = a, s_red[1][threadIdx.x] = b;
  __syncthreads();
  for (uint32_t o = 128; o > 0; o >>= 1) {
    if (threadIdx.x < o) {
      s_red[0][threadIdx.x] = max(s_red[0][threadIdx.x], s_red[0][threadIdx.x + o]);
      s_red[1][threadIdx.x] = max(s_red[1][threadIdx.x], s_red[1][threadIdx.x + o]);
    }
    __syncthreads();
  }
  if (threadIdx.x == 0) {
    const bool ok = mark[0] == 0u && fail[0] == 0u && window_safe(s_red[0][0], s_red[1][0]);
    verdict[0] = ok ? 0u : 3u, verdict[1] = 0u;
    if (ok) mark[0] = 2u;
  }
}

// ---- i8 operand path -----------------------------------------------------------------------------------
// A sketch HV is hv[d] = 2*count[d] - n (src/hd.rs:29,84-87): all entries of a row have the parity e = n & 1, so
//     x = 2*c - e,   c = (x + e) >> 1   (exact; c is the bit count centred on n/2, sigma = sqrt(n)/2),
//     dot(r, q) = 4*sum c_r*c_q - 2*e_q*S_r - 2*e_r*S_q + D*e_r*e_q,     S = sum_d c[d].
// For sketches of up to ~3 500 hashes (genomes up to ~5 Mbp at scaled = 1500) c fits a signed byte for all but a
// ~1e-5 fraction of the entries, so G = sum a_r*a_q (a = c clamped to [-127, 127]) runs on
// v_mfma_i32_16x16x64_i8: twice the K per instruction AND half the operand bytes of the f16 path (the kernel is
// co-limited by the L2 -> LDS feed), exact in the i32 accumulator without any window logic.  The few clamped
// entries ("outliers", residual b = c - a) are repaired exactly, outside the GEMM:
//     sum c_i*c_j = G + sum_{d in out(i)} b_i[d]*c_j[d] + sum_{d in out(j)} a_i[d]*b_j[d]
// Both sums are evaluated in the epilogue, only for the few candidates that survive the threshold pre-filter AND sit
// in a row / column that has clamped entries (~4 % of the rows): a row's entries (dim, b) are consecutive in a sorted
// list, c_j[d] and a_i[d] are read back from the original i16 matrices.  Rows of mixed
// parity, residuals beyond a byte, a row with more than 255 clamped entries or an overflow of the entry list veto
// the path on the device and the f16 kernels queued behind it run instead; the dot product is the same integer
// either way.
constexpr uint32_t I8_ROW_ENT_MAX = 256;  // clamped entries of a row the prepass looks at (the slot word counts to 255)
// Every row owns I8_ROW_SLOTS consecutive entries of the list (row r of side s at (s ? R : 0) * SLOTS + r * SLOTS): no
// reservation at all.  (Until round 3 the rows appended to one compact list through ONE atomic counter: the same-address
// atomics of 10 000 rows serialise at ~9 ns, and the prepass of sketches with an entry in every row -- 4 500 hashes and
// more -- took 0.075-0.135 ms instead of 0.03.)  A row with more entries than slots vetoes the i8 path for the call:
// at 16 slots that is one row in 10^5 at 6 000 hashes (4.2 entries per row on average), every call at 7 000.
constexpr uint32_t I8_ROW_SLOTS = 16;
struct I8Outlier {
  uint32_t row;
  uint16_t d;
  int8_t b;
  uint8_t side;  // 0 = reference matrix, 1 = query matrix
};
// ctrl words (device): [0] outlier count, [1] failure bits, [2] phase-0 slack of the GEMM epilogue, [3] entries of side 0,
//                      [4] verdict (1 = i8 path valid), [5] K-steps of 128 bytes
// A row's control record as it travels between GPUs (hg_dist_prep_ops_dev -> hg_dist_block_ops_dev): what the rank that owns
// the row computed for it, 72 bytes against the row's 4 KiB of byte operands
struct I8RowMeta {
  int32_t info;                 // 2 * S + e
  int32_t slot;                 // entries (8 bits) << 14 | sum |b| (14 bits); 0 = none
  uint32_t ent[I8_ROW_SLOTS];   // the clamped entries: d | (uint8)b << 16
};
static_assert(sizeof(I8RowMeta) == 72, "hg_dist_ops_meta_bytes");
// meta != nullptr: the per-row words go into packed records instead of the rowinfo / rowslot / rowfirst / list arrays
__global__ __launch_bounds__(256) void prep_i8_kernel(const int16_t *__restrict__ hv, uint32_t rows, uint32_t hv_d,
                                                      uint32_t kp8, uint32_t ldk8, int8_t *__restrict__ out_a,
                                                      int32_t *__restrict__ rowinfo, int32_t *__restrict__ rowslot,
                                                      uint32_t *__restrict__ rowfirst, I8Outlier *__restrict__ list,
                                                      uint32_t list_base, uint32_t *__restrict__ ctrl, uint32_t side,
                                                      I8RowMeta *__restrict__ meta = nullptr) {
  // the row's clamped entries are collected in LDS (one wave = one row) and go to the global list as ONE contiguous
  // range reserved with a single atomic: no sort, no second kernel, and the list can be as long as memory allows
  __shared__ uint32_t s_ent[4][I8_ROW_ENT_MAX];
  __shared__ uint32_t s_n[4];
  const uint32_t lane = threadIdx.x & 63, wv = threadIdx.x >> 6, row = blockIdx.x * 4 + wv;
  if (row >= rows) return;  // whole wave
  if (lane == 0) s_n[wv] = 0;
  __builtin_amdgcn_wave_barrier();
  const int16_t *__restrict__ src = hv + (size_t)row * hv_d;
  const int32_t x0 = src[0], e = x0 & 1;
  int32_t S = 0;
  uint32_t par = 0, bad = 0;
  const bool vec_ok = (hv_d % 8 == 0) && ((reinterpret_cast<uintptr_t>(src) & 15) == 0);
  const uint32_t nchunks = (kp8 + 511) / 512;
  for (uint32_t q0 = 0; q0 < nchunks; q0 += 4) {  // four 16-byte loads in flight per lane
    uint32_t d0s[4];
    uint4 raw[4];
    bool vec[4];
#pragma unroll
    for (int t = 0; t < 4; ++t) {
      // chunk order rotated by the row index: a power-of-two row pitch otherwise sends every wave to the same channels
      d0s[t] = q0 + t < nchunks ? ((q0 + t + row) % nchunks) * 512 + lane * 8 : kp8;
      vec[t] = vec_ok && d0s[t] + 8 <= hv_d;
      raw[t] = make_uint4(0, 0, 0, 0);
      if (vec[t]) raw[t] = *reinterpret_cast<const uint4 *>(src + d0s[t]);
    }
#pragma unroll
    for (int t = 0; t < 4; ++t) {
      const uint32_t d0 = d0s[t];
      if (d0 >= kp8) continue;
      int32_t x[8];
      if (vec[t]) {
        const uint32_t w[4] = {raw[t].x, raw[t].y, raw[t].z, raw[t].w};
#pragma unroll
        for (int i = 0; i < 4; ++i) x[2 * i] = (int16_t)(w[i] & 0xffffu), x[2 * i + 1] = (int16_t)(w[i] >> 16);
      } else {
#pragma unroll
        for (int i = 0; i < 8; ++i) x[i] = d0 + i < hv_d ? (int32_t)src[d0 + i] : -e;  // padding: c = 0
      }
      uint32_t pk[2] = {0, 0};
      const bool all_real = d0 + 8 <= hv_d;
#pragma unroll
      for (int i = 0; i < 8; ++i) {
        par |= (all_real || d0 + i < hv_d) ? (uint32_t)((x[i] ^ x0) & 1) : 0u;
        const int32_t cc = (x[i] + e) >> 1;
        S += cc;
        const int32_t a = cc > 127 ? 127 : (cc < -127 ? -127 : cc), b = cc - a;
        if (b != 0) {
          if (b > 127 || b < -127) bad |= 4u;
          const uint32_t idx = atomicAdd(&s_n[wv], 1u);
          if (idx < I8_ROW_ENT_MAX) s_ent[wv][idx] = (d0 + i) | ((uint32_t)(uint8_t)(int8_t)b << 16);
        }
        pk[i >> 2] |= (uint32_t)(uint8_t)(int8_t)a << (8 * (i & 3));
      }
      *reinterpret_cast<uint2 *>(out_a + (size_t)row * ldk8 + d0) = make_uint2(pk[0], pk[1]);
    }
  }
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) S += __shfl_xor(S, o);
  __builtin_amdgcn_wave_barrier();
  asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
  const uint32_t n_raw = s_n[wv], n = n_raw < I8_ROW_ENT_MAX ? n_raw : I8_ROW_ENT_MAX;
  const uint32_t n_st = n < I8_ROW_SLOTS ? n : I8_ROW_SLOTS;  // entries stored
  uint32_t bs = 0;  // sum |b| over the row's entries (the epilogue's per-row slack)
  for (uint32_t t = lane; t < n; t += 64) {
    const int32_t bb = (int8_t)(uint8_t)(s_ent[wv][t] >> 16);
    bs += (uint32_t)(bb < 0 ? -bb : bb);
  }
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) bs += __shfl_xor(bs, o);
  if (n_raw > I8_ROW_SLOTS || bs >= (1u << 14)) bad |= 2u;  // more than the row's slots / the slot word can describe
  const uint32_t base = list_base + row * I8_ROW_SLOTS;
  const int32_t slotw = n_st ? (int32_t)(((n_st & 255u) << 14) | (bs & 0x3fffu)) : 0;  // entries (8 bits) | sum |b| (14 bits); 0 = none
  if (meta) {
    if (lane < I8_ROW_SLOTS) meta[row].ent[lane] = lane < n_st ? (s_ent[wv][lane] & 0x00FFFFFFu) : 0u;
    if (lane == 0) meta[row].info = 2 * S + e, meta[row].slot = slotw;
  } else if (lane < n_st) {
    const uint32_t v = s_ent[wv][lane];
    list[base + lane] = I8Outlier{row, (uint16_t)(v & 0xffffu), (int8_t)(uint8_t)(v >> 16), (uint8_t)side};
  }
  const bool anypar = __any(par != 0), anybad4 = __any((bad & 4u) != 0), anybad2 = __any((bad & 2u) != 0);
  if (lane == 0) {
    if (!meta) {
      rowinfo[row] = 2 * S + e;
      rowfirst[row] = base;
      rowslot[row] = slotw;
    }
    const uint32_t fl = (anypar ? 1u : 0u) | (anybad2 ? 2u : 0u) | (anybad4 ? 4u : 0u);
    if (fl) atomicOr(&ctrl[1], fl);
  }
}

// gathered records -> the arrays the GEMM's epilogue reads (side 0); workgroup 0 also folds the owners' failure flags into
// the call's control words
__global__ __launch_bounds__(256) void unpack_meta_kernel(const I8RowMeta *__restrict__ meta, uint32_t rows, int32_t *__restrict__ rowinfo,
                                                          int32_t *__restrict__ rowslot, uint32_t *__restrict__ rowfirst,
                                                          I8Outlier *__restrict__ list, const uint32_t *__restrict__ flags,
                                                          uint32_t n_flags, uint32_t *__restrict__ ctrl) {
  if (blockIdx.x == 0) {
    uint32_t fl = 0;
    for (uint32_t i = threadIdx.x; i < n_flags; i += blockDim.x) fl |= flags[i];
    if (fl) atomicOr(&ctrl[1], fl);
  }
  const uint32_t t = blockIdx.x * blockDim.x + threadIdx.x, row = t / I8_ROW_SLOTS, k = t % I8_ROW_SLOTS;
  if (row >= rows) return;
  const I8RowMeta &m = meta[row];
  const uint32_t n_st = ((uint32_t)m.slot >> 14) & 255u;
  if (k == 0) rowinfo[row] = m.info, rowslot[row] = m.slot, rowfirst[row] = row * I8_ROW_SLOTS;
  if (k < n_st) {
    const uint32_t v = m.ent[k];
    list[row * I8_ROW_SLOTS + k] = I8Outlier{row, (uint16_t)(v & 0xffffu), (int8_t)(uint8_t)(v >> 16), (uint8_t)0};
  }
}

// The i8 attempt is valid iff no row broke the scheme (ctrl[1]: parity / residual / per-row limits, among them the
// row's list slots).  Every workgroup of the GEMM evaluates this by itself.
__device__ __forceinline__ bool i8_attempt_valid(const uint32_t *ctrl, uint32_t) {
  return ctrl[1] == 0u;
}


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
__device__ unsigned long long g_dist_tile_all[2048][5];  // per workgroup: entry, main loop done, tile done, candidates evaluated, XCC id | HW_ID << 8
#define HG_TSTAMP(pt)                                                                                      \
  if ((threadIdx.x & 63) == 0 && blockIdx.x >= 512 && blockIdx.x < 528)                                    \
    g_dist_tile_stamps[blockIdx.x - 512][threadIdx.x >> 6][pt] = __builtin_amdgcn_s_memtime();            \
  if (threadIdx.x == 0 && blockIdx.x < 2048 && ((pt) == 0 || (pt) == 2 || (pt) == 5))                      \
    g_dist_tile_all[blockIdx.x][(pt) == 0 ? 0 : ((pt) == 2 ? 1 : 2)] = __builtin_amdgcn_s_memtime();       \
  if (threadIdx.x == 0 && blockIdx.x < 2048 && ((pt) == 1 || (pt) == 2))                                   \
    g_dist_tile_real[blockIdx.x][(pt)-1] = __builtin_amdgcn_s_memrealtime();
#else
#define HG_STAMP(pt)
#define HG_TSTAMP(pt)
#endif
// FULL: every ANI is evaluated and stored (parity / small problems).  Otherwise only pairs that can
// reach ani_th are evaluated: one multiply-compare rejects the rest (ANI is monotone in the Jaccard
// index), the exact reference arithmetic decides the survivors.
// GLDS (big geometry only): operand tiles go HBM -> LDS by LDS-DMA (global_load_lds, 16 B per lane, no
// VGPR staging and no ds_write pass).  The DMA writes each wave-instruction's 1 KiB linearly, so the
// LDS image is unpadded [row][8 chunks of 16 B] and bank conflicts are removed by an XOR swizzle of the
// chunk index with (row >> 1) & 7 -- applied to the per-lane SOURCE address when loading and to the
// fragment address when reading (same involution on both sides).
typedef int int4v __attribute__((ext_vector_type(4)));
typedef int int8v __attribute__((ext_vector_type(8)));
template <int... Js, class F>
__device__ __forceinline__ void dist_static_for(std::integer_sequence<int, Js...>, F &&f) {
  (f(std::integral_constant<int, Js>{}), ...);
}
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
  constexpr bool CENT = (I8 && !HAM) || CEN;  // the epilogue works on centred counts: info words, dot = 4 G - ...
  static_assert(!I8 || (GLDS && !CHUNKED && !FULL), "the i8 operand path exists for the thresholded LDS-DMA geometries");
  static_assert(!HAM || I8, "the Hamming epilogue rides on the i8 operand path");
  static_assert(!FP4 || HAM, "e2m1 operands exist for the Hamming search only");
  HG_TSTAMP(0)
#ifdef HG_DIST_STAMPS
  if (threadIdx.x == 0 && blockIdx.x < 2048)
    g_dist_tile_all[blockIdx.x][3] = 0, g_dist_tile_all[blockIdx.x][1] = 0, g_dist_tile_all[blockIdx.x][2] = 0,
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
  constexpr int LROW = GLDS ? BK : LDS_ROW;  // elements per LDS row
  constexpr int BM = TC::BM, BN = TC::BN, WTM = TC::WTM, NWN = TC::NWN, THREADS = TC::THREADS, LOADS = TC::LOADS;
  static_assert(GLDS || (LOADS == 4 && TC::LOADS_B == 4), "staging macros move 4 pieces per operand");
  static_assert(NT == 4 || GLDS, "wide tiles exist for the LDS-DMA variant only");
  // two LDS stages of (A tile + B tile)
  extern __shared__ __attribute__((aligned(16))) _Float16 sAB[];
  constexpr uint32_t A_ELEMS = BM * LROW, B_ELEMS = BN * LROW;
  constexpr uint32_t TILE_ELEMS = A_ELEMS;           // offset of the B tile inside a stage
  constexpr uint32_t STAGE_ELEMS = A_ELEMS + B_ELEMS;
  constexpr uint32_t SROWS = THREADS / 8;            // rows covered by one staging pass

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

  const uint32_t tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const uint32_t wm = wave / NWN, wn = wave % NWN;  // 2 x NWN waves, (WTM*16) x 64 each
  const uint32_t fr = lane & 15, fq = lane >> 4;

  typedef typename std::conditional<I8 && !FP4, int4v, float4v>::type acc_t;  // i8 operands accumulate in exact i32
  acc_t acc[WTM][NT];
  int32_t iacc[CHUNKED ? WTM : 1][CHUNKED ? NT : 1][4];
#pragma unroll
  for (int m = 0; m < WTM; ++m)
#pragma unroll
    for (int n = 0; n < NT; ++n) acc[m][n] = acc_t{};
  if (CHUNKED) {
#pragma unroll
    for (int m = 0; m < WTM; ++m)
#pragma unroll
      for (int n = 0; n < NT; ++n)
#pragma unroll
        for (int r = 0; r < 4; ++r) iacc[CHUNKED ? m : 0][CHUNKED ? n : 0][r] = 0;
  }

  // staging: thread t moves 4 x 16 B of A and of B per K-step: row = t/8 + SROWS*i, 16-byte piece t%8
  const uint32_t srow = tid >> 3, spc = tid & 7;
  const _Float16 *gA = g.A + (size_t)(row0 + srow) * g.ldk + spc * 8;
  const _Float16 *gB = g.B + (size_t)(col0 + srow) * g.ldk + spc * 8;
  const size_t rstep = (size_t)SROWS * g.ldk;
  const uint32_t st_off = srow * LROW + spc * 8;                    // this thread's slot in a tile
  // fragment bases in a stage; with the swizzle the lane's 16-byte chunk is (kk*4 + fq) ^ ((row>>1)&7),
  // and (row>>1)&7 == (fr>>1)&7 because all row bases are multiples of 16
  const uint32_t swz = (fr >> 1) & 7;
  const uint32_t fa_off = (wm * WTM * 16 + fr) * LROW + (GLDS ? (fq ^ swz) * 8 : fq * 8);
  const uint32_t fb_off = TILE_ELEMS + (wn * (NT * 16) + fr) * LROW + (GLDS ? (fq ^ swz) * 8 : fq * 8);
  // kk = 1 adds 4 chunks: (4 + fq) ^ swz = (fq ^ swz) ^ 4
  const int32_t kk1_off = GLDS ? ((((fq ^ swz) ^ 4) - (int32_t)(fq ^ swz)) * 8) : 32;
  uint4 ra0, ra1, ra2, ra3, rb0, rb1, rb2, rb3;
#define HG_GLOAD(k0)                                                    \
  ra0 = *reinterpret_cast<const uint4 *>(gA + (k0));                    \
  ra1 = *reinterpret_cast<const uint4 *>(gA + rstep + (k0));            \
  ra2 = *reinterpret_cast<const uint4 *>(gA + 2 * rstep + (k0));        \
  ra3 = *reinterpret_cast<const uint4 *>(gA + 3 * rstep + (k0));        \
  rb0 = *reinterpret_cast<const uint4 *>(gB + (k0));                    \
  rb1 = *reinterpret_cast<const uint4 *>(gB + rstep + (k0));            \
  rb2 = *reinterpret_cast<const uint4 *>(gB + 2 * rstep + (k0));        \
  rb3 = *reinterpret_cast<const uint4 *>(gB + 3 * rstep + (k0));
#define HG_LSTORE(stage)                                                                   \
  {                                                                                        \
    _Float16 *lA = sAB + (stage) * STAGE_ELEMS + st_off, *lB = lA + TILE_ELEMS;            \
    *reinterpret_cast<uint4 *>(lA) = ra0;                                                  \
    *reinterpret_cast<uint4 *>(lA + SROWS * LROW) = ra1;                                \
    *reinterpret_cast<uint4 *>(lA + 2 * SROWS * LROW) = ra2;                            \
    *reinterpret_cast<uint4 *>(lA + 3 * SROWS * LROW) = ra3;                            \
    *reinterpret_cast<uint4 *>(lB) = rb0;                                                  \
    *reinterpret_cast<uint4 *>(lB + SROWS * LROW) = rb1;                                \
    *reinterpret_cast<uint4 *>(lB + 2 * SROWS * LROW) = rb2;                            \
    *reinterpret_cast<uint4 *>(lB + 3 * SROWS * LROW) = rb3;                            \
  }

  // Software pipeline with ONE barrier per K-step.  A step is PHASES phases of 8 MFMAs; the fragments of
  // phase t+1 are read from LDS while phase t multiplies.  The barrier sits BEFORE the last phase of a
  // step, not after it: at that point every fragment of the current stage is already in registers, so
  // once all waves have arrived (and, DMA variant, the next tile has landed: vmcnt(0)) the stage can be
  // refilled and the first fragments of the next stage can be read -- both under the cover of the 8 MFMAs
  // still to issue, instead of an idle matrix pipe right after every barrier.
  //   register-staged: top of step k stores tile k+1 (requested during step k-1) into stage (k+1)&1 and
  //                    requests tile k+2; the barrier in the last phase publishes it.
  //   DMA            : right after the barrier of step k the DMA of tile k+2 starts into stage k&1 (a whole
  //                    step of latency cover).
  const uint32_t nsteps = g.Kp / BK;
  // LDS-DMA staging: thread t fills slots s = i*THREADS + t (i < 4) of each operand tile; slot s is
  // row s/8, LDS chunk s%8, and holds global chunk (s%8) ^ ((row>>1)&7) of that row.  The wave's 64 slots
  // of one instruction are 1 KiB contiguous in LDS, as the DMA requires.
  // The DMA is issued as buffer_load_dwordx4 ... lds through a per-workgroup buffer descriptor (base = the
  // tile's first row, 32-bit per-lane offset, K offset in an SGPR).  The global_load_lds form moves the same
  // bytes, but being FLAT-encoded it makes the compiler flush lgkmcnt to 0 at every LDS dependency while
  // one is in flight, which serialises the fragment reads below with the MFMAs.
  // DMA issue is left to LW of the 8 waves -- one per SIMD when LW = 4: a wave whose VMEM instructions queue
  // up behind the workgroup's burst cannot issue MFMAs meanwhile, and with every wave loading right after the
  // barrier both waves of a SIMD sit in that queue together while the matrix pipe idles.  With one loader
  // per SIMD its partner keeps the pipe busy and the loader catches up while the partner waits at the barrier.
  constexpr int HG_DMA_LOADER_WAVES = 4;
  constexpr int LW = HG_DMA_LOADER_WAVES < THREADS / 64 ? HG_DMA_LOADER_WAVES : THREADS / 64, LT = LW * 64;  // loader waves / threads
  constexpr int PA = BM * 8 / LT, PB = BN * 8 / LT;      // 16-byte pieces per loader thread, A / B tile
  // byte offsets of this thread's pieces inside the A / B row block (fixed-size arrays: a template-sized
  // array here makes hipcc drop the kernel's host stub without a diagnostic)
  uint32_t vA[8], vB[10];
  static_assert(!GLDS || (PA <= 8 && PB <= 10), "piece tables too small");
  __amdgpu_buffer_rsrc_t rsA, rsB;
  if (GLDS) {
    constexpr int PMAX = PA > PB ? PA : PB;
#pragma unroll
    for (int i = 0; i < PMAX; ++i) {
      const uint32_t sl = i * LT + (tid & (LT - 1)), r = sl >> 3, ch = (sl & 7) ^ ((r >> 1) & 7);
      const uint32_t off = (r * g.ldk + ch * 8) * 2;
      if (i < PB) vB[i] = off;
      if (i < PA) vA[i] = off;
    }
    rsA = __builtin_amdgcn_make_buffer_rsrc(const_cast<_Float16 *>(g.A + (size_t)(HG_EXP(512) ? 0u : row0) * g.ldk), 0, 0x7fffffff, 0x00020000);
    rsB = __builtin_amdgcn_make_buffer_rsrc(const_cast<_Float16 *>(g.B + (size_t)(HG_EXP(512) ? 0u : col0) * g.ldk), 0, 0x7fffffff, 0x00020000);
  }
  typedef __attribute__((address_space(3))) void *lds_ptr_t;
#define HG_DMA(stage, k0)                                                                                   \
  {                                                                                                         \
  if (wave < (uint32_t)LW) {                                                                                \
    _Float16 *wbase = sAB + (stage) * STAGE_ELEMS + wave * 64 * 8; /* this wave's 1 KiB of instruction 0 */ \
    _Pragma("unroll") for (int i = 0; i < (PA > PB ? PA : PB); ++i) {                                       \
      if (i < PA)                                                                                           \
        __builtin_amdgcn_raw_ptr_buffer_load_lds(rsA, (lds_ptr_t)(wbase + i * LT * 8), 16, vA[i < PA ? i : 0], (HG_EXP(512) ? (k0) & 1023u : (k0)) * 2, 0, 0); \
      if (i < PB)                                                                                           \
        __builtin_amdgcn_raw_ptr_buffer_load_lds(rsB, (lds_ptr_t)(wbase + TILE_ELEMS + i * LT * 8), 16, vB[i < PB ? i : 0], (HG_EXP(512) ? (k0) & 1023u : (k0)) * 2, 0, 0); \
    }                                                                                                       \
  }                                                                                                         \
  }
  // A fragments per phase: 2 (x NT B fragments = 8..10 MFMAs, the other wave of the SIMD covers the fragment latency)
  constexpr int AF = 2;
  const int32_t fp4_unit_scale = 0x7f7f7f7f;  // FP4: E8M0 block scales of 2^0 for every 32-element block
  constexpr int MP = WTM / AF, PHASES = (BK / 32) * MP;
  half8 bfr[2][NT], afr[2][AF];
  // fragments of phase (kk, mp) of the stage whose fragment bases are pa / pb, into buffer set `buf`
#define HG_FRAGS(buf, pa, pb, kk, mp)                                                                       \
  {                                                                                                         \
    const int32_t ko_ = (kk) ? kk1_off : 0;                                                                 \
    if ((mp) == 0) {                                                                                        \
      _Pragma("unroll") for (int n = 0; n < NT; ++n)                                                         \
          bfr[(kk) & 1][n] = *reinterpret_cast<const half8 *>((pb) + n * 16 * LROW + ko_);                  \
    }                                                                                                       \
    _Pragma("unroll") for (int i_ = 0; i_ < AF; ++i_)                                                       \
        afr[buf][i_] = *reinterpret_cast<const half8 *>((pa) + (AF * (mp) + i_) * 16 * LROW + ko_);         \
  }
  // The tile's row / column words for the epilogue are staged NOW: their global loads are issued in front of the
  // first operand tile's, travel with it, and the barrier below publishes what is computed from them (fetched after
  // the K loop they cost a dependent-load latency per tile with nothing to hide it behind).  Per row / column: the
  // norm, the i8 path's info / outlier words, and the phase-0 threshold (see the epilogue), so that the accumulator
  // sweep reads ONE float per row and column.
  int32_t *s_nr = reinterpret_cast<int32_t *>(reinterpret_cast<char *>(sAB) + dist_lds_main_bytes<BIG, NT, GLDS>()), *s_nq = s_nr + BM;
  int32_t *s_ir = s_nq + BN, *s_iq = s_ir + BM;                                 // i8 path: 2*S + e per row / column
  int32_t *s_sr = s_iq + BN, *s_sq = s_sr + BM;                                 // ... the outlier-entry slots
  uint32_t *s_fr = reinterpret_cast<uint32_t *>(s_sq + BN), *s_fq = s_fr + BM;  // ... and their first entries
  float *s_ur = reinterpret_cast<float *>(s_fq + BN), *s_tq = s_ur + BM;        // phase-0 thresholds
  uint32_t *s_er = reinterpret_cast<uint32_t *>(s_tq + BN), *s_eq = s_er + BM;  // i8 path: the first clamped entry itself, d | b << 16
  uint32_t *s_cnt = s_eq + BN;                                                  // per-wave hit counts + the workgroup's base
  uint32_t *s_tot = s_cnt + 2 * (THREADS / 64) + 4, *s_fill = s_tot + THREADS / 64;  // (behind the list lengths and flags) candidates per wave: counted / appended
  constexpr int32_t NORM_SAFE = 1 << 29;
  constexpr int WORD_PASSES = (BM + BN + THREADS - 1) / THREADS;
  int32_t w_nv[WORD_PASSES], w_info[WORD_PASSES], w_slot[WORD_PASSES];
  uint32_t w_first[WORD_PASSES];
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
  if (GLDS) {
    HG_DMA(0, 0)
  } else {
    HG_GLOAD(0)
    HG_LSTORE(0)
    if (nsteps > 1) {
      HG_GLOAD(BK)
    }
  }
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
  __syncthreads();  // (with DMA in flight hipcc's barrier also waits vmcnt(0): stage 0 has landed)
  HG_TSTAMP(1)
  if (GLDS && nsteps > 1) HG_DMA(1, BK)
  if (!HG_EXP(2)) HG_FRAGS(0, sAB + fa_off, sAB + fb_off, 0, 0)
  uint32_t in_chunk = 0;
  for (uint32_t ks = 0; ks < nsteps; ++ks) {
    const uint32_t cur = ks & 1;
    HG_STAMP(0)
    if (!GLDS && ks + 1 < nsteps) {
      HG_LSTORE(cur ^ 1)
      if (ks + 2 < nsteps) {
        const uint32_t k2 = (ks + 2) * BK;
        HG_GLOAD(k2)
      }
    }
    const _Float16 *fA = sAB + cur * STAGE_ELEMS + fa_off, *fB = sAB + cur * STAGE_ELEMS + fb_off;
    const _Float16 *nA = sAB + (cur ^ 1) * STAGE_ELEMS + fa_off, *nB = sAB + (cur ^ 1) * STAGE_ELEMS + fb_off;
#pragma unroll
    for (int t = 0; t < PHASES; ++t) {
      const int kk = t / MP, mp = t % MP;
      if (t + 1 < PHASES) {
        if (!HG_EXP(2) && !(HG_EXP(32) && ks)) HG_FRAGS((t + 1) & 1, fA, fB, (t + 1) / MP, (t + 1) % MP)
      } else {
        // every fragment read of this stage must have returned before another wave may refill it
        HG_STAMP(1)
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        HG_STAMP(2)
        if (!HG_EXP(16)) __syncthreads();
        HG_STAMP(3)
        if (GLDS && ks + 2 < nsteps && !HG_EXP(1)) HG_DMA(cur, (ks + 2) * BK)
        HG_STAMP(4)
        if (ks + 1 < nsteps && !HG_EXP(2) && !HG_EXP(32)) HG_FRAGS(0, nA, nB, 0, 0)
      }
      __builtin_amdgcn_sched_barrier(0);
      if (HG_EXP(8)) {  // fragment reads without the MFMAs
#pragma unroll
        for (int i = 0; i < AF; ++i) asm volatile("" ::"v"(afr[t & 1][i]));
#pragma unroll
        for (int n = 0; n < NT; ++n) asm volatile("" ::"v"(bfr[kk & 1][n]));
      } else if (!HG_EXP(2)) {
#pragma unroll
        for (int i = 0; i < AF; ++i)
#pragma unroll
          for (int n = 0; n < NT; ++n)
            if constexpr (FP4) {  // 32 e2m1 values per lane in the fragment's 16 bytes; block scales 2^0 (E8M0 127).
              // Written as asm: the builtin takes 8-register operand vectors (the e4m3 width); padding the 16-byte
              // fragments to that width costs a copy of every fragment per K-step and 60 VGPRs (spills at 256).
              asm volatile("v_mfma_scale_f32_16x16x128_f8f6f4 %0, %1, %2, %0, %3, %3 op_sel_hi:[0,0,0] cbsz:4 blgp:4"
                           : "+v"(acc[AF * mp + i][n])
                           : "v"(__builtin_bit_cast(int4v, afr[t & 1][i])), "v"(__builtin_bit_cast(int4v, bfr[kk & 1][n])),
                             "v"(fp4_unit_scale));
            } else if constexpr (I8)  // the same 16-byte fragments hold 16 k-consecutive bytes per lane: one instruction covers K = 64
              acc[AF * mp + i][n] = __builtin_amdgcn_mfma_i32_16x16x64_i8(__builtin_bit_cast(int4v, afr[t & 1][i]),
                                                                          __builtin_bit_cast(int4v, bfr[kk & 1][n]),
                                                                          acc[AF * mp + i][n], 0, 0, 0);
            else
              acc[AF * mp + i][n] =
                  __builtin_amdgcn_mfma_f32_16x16x32_f16(afr[t & 1][i], bfr[kk & 1][n], acc[AF * mp + i][n], 0, 0, 0);
      }
      __builtin_amdgcn_sched_barrier(0);
    }
    HG_STAMP(5)
    if (CHUNKED && ++in_chunk == g.chunk_steps) {  // move the exact f32 partial sums into i32
      in_chunk = 0;
#pragma unroll
      for (int m = 0; m < WTM; ++m)
#pragma unroll
        for (int n = 0; n < NT; ++n) {
#pragma unroll
          for (int r = 0; r < 4; ++r) iacc[CHUNKED ? m : 0][CHUNKED ? n : 0][r] += (int32_t)acc[m][n][r];
          acc[m][n] = acc_t{};
        }
    }
  }
  HG_TSTAMP(2)
  if constexpr (FP4)  // asm MFMAs: the hazard recogniser does not know that the accumulators come from the matrix pipe
    asm volatile("s_nop 7\n\ts_nop 7\n\ts_nop 7" ::: "memory");
  // The epilogue reuses the operand stages for its per-wave candidate lists (every fragment read was retired by the
  // last in-loop barrier); phase 2 gathers the norms staged at kernel entry by candidate (from global memory each
  // 64-candidate batch paid a full dependent-load latency: 0.12 ms per launch at 1.3 M hits).
#undef HG_GLOAD
#undef HG_LSTORE
#undef HG_DMA
#undef HG_FRAGS

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
    if (tid == 0) {
      uint32_t total = 0;
#pragma unroll
      for (uint32_t w = 0; w < NW_; ++w) total += s_cnt[w];
      s_cnt[NW_] = total ? atomicAdd(g.hit_count, total) : 0u;
    }
    __syncthreads();
    HG_TSTAMP(7)
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
    staged = 0;
    __syncthreads();  // the lists may be refilled only after every wave has read them
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
  //  * LANE MASKS (every thresholded kernel, tiles off the diagonal of a symmetric comparison -- there a candidate is
  //    "passes phase 0"; the f16 kernels' denominator test of the slab path is only a cheaper filter in front of the
  //    exact phase 2: 0.69 -> 0.61 ms at 10 000 x 10 000 without it, the windowed kernel 0.85 -> 0.78): every lane shifts the sign of `d - ur - tq` of its 4 * NT elements of a 16-row slab
  //    into one mask word per slab, no branches.  One barrier makes the waves' candidate counts known to all: a tile
  //    without candidates ends there; if no list can overflow, every wave then appends its candidates on its own --
  //    per slab one LDS atomic per lane that has any reserves its run of the list, predicated stores fill it -- and
  //    the workgroup meets again in flush_all.
  //  * SLABS (the full-matrix mode, diagonal tiles of a symmetric comparison, tiles whose candidates may overflow a
  //    list): per 16-row slab the 4 * NT compares are OR-ed on the scalar side into one wave-uniform branch; a slab
  //    with candidates takes one ballot per element, and a barrier per slab makes the decision to empty the lists
  //    uniform.
  constexpr bool LANE_MASKS = !FULL;  // (f16 operands: phase 2 is exact, the slab path's denominator test is only a cheaper filter)
  constexpr uint32_t BNC_LANE = 80, BNC_WAVE = 64 * BNC_LANE;  // bytes of a lane's / a wave's bounce buffer (append loop)
  static_assert(NT * 16 <= (int)BNC_LANE, "a lane's slab fits its bounce buffer");
  constexpr uint32_t SLAB_BITS = (1u << (4 * NT)) - 1u;
  bool by_lane = false, have_masks = false;  // workgroup-uniform
  uint32_t notpass[LANE_MASKS ? WTM : 1], lane_cands = 0, wave_cands = 0;  // (wave_cands: lane w holds wave w's count)
  if constexpr (LANE_MASKS) {
    if (!(g.symmetric && row0 + g.ref_off + (uint32_t)BM - 1u >= col0 + g.qry_off)) {
      have_masks = true;
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
        notpass[m] = np;
        lane_total += (uint32_t)__popc(~np & SLAB_BITS);
      });
      lane_cands = lane_total;
      for (int o = 32; o > 0; o >>= 1) lane_total += __shfl_xor(lane_total, o);
      if (lane == 0) s_tot[wave] = lane_total;
      __syncthreads();
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
extern "C" int hg_debug_dist_tile_all(unsigned long long *out /* 2048 * 5 */) {
  return (int)hipMemcpyFromSymbol(out, HIP_SYMBOL(g_dist_tile_all), sizeof(unsigned long long) * 2048 * 5);
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
