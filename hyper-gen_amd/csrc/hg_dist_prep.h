// hg_dist_prep.h -- operand prepasses of the ANI GEMM: i16 hypervectors -> f16 / centred f16 / centred i8 operands, with the
// statistics that prove each form exact (see hg_dist_kernels.hip for the scheme).  Private to hg_dist_kernels.hip.
#pragma once
#include "hg_dist_common.h"

namespace {

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
  constexpr int INFL = 4;  // 16-byte loads in flight per lane (8 -- a whole row of 4 096 dimensions -- costs registers: 27 -> 30 us for 10 000 rows)
  for (uint32_t q0 = 0; q0 < nchunks; q0 += INFL) {
    uint32_t d0s[INFL];
    uint4 raw[INFL];
    bool vec[INFL];
#pragma unroll
    for (int t = 0; t < INFL; ++t) {
      // chunk order rotated by the row index: a power-of-two row pitch otherwise sends every wave to the same channels
      d0s[t] = q0 + t < nchunks ? ((q0 + t + row) % nchunks) * 512 + lane * 8 : kp8;
      vec[t] = vec_ok && d0s[t] + 8 <= hv_d;
      raw[t] = make_uint4(0, 0, 0, 0);
      if (vec[t]) raw[t] = *reinterpret_cast<const uint4 *>(src + d0s[t]);
    }
#pragma unroll
    for (int t = 0; t < INFL; ++t) {
      const uint32_t d0 = d0s[t];
      if (d0 >= kp8) continue;
      if (vec[t]) {
        // Eight real values as four dwords: packed 16-bit arithmetic, two values per instruction (v_pk_add_u16, v_pk_ashrrev_i16,
        // v_pk_max_i16 / v_pk_min_i16, v_pk_sub_i16, v_dot2c_i32_i16), one test for clamped entries per chunk.  (Element by
        // element this loop was ~19 instructions per value and the kernel, 10 waves per SIMD deep, was bound by them: 31 us for
        // 10 000 rows of which ~20 were issue time.)  x + e may wrap at 32 767: the residual then exceeds a byte and the row
        // vetoes the path, which is what the true value does as well.
        const uint32_t w[4] = {raw[t].x, raw[t].y, raw[t].z, raw[t].w};
        const short2v ev = {(short)e, (short)e}, hi = {127, 127}, lo = {-127, -127}, one = {1, 1};
        const uint32_t x0p = (uint32_t)(uint16_t)x0 * 0x00010001u;
        uint32_t ab[4], ball = 0;
        short2v bv[4];
#pragma unroll
        for (int i = 0; i < 4; ++i) {
          par |= (w[i] ^ x0p) & 0x00010001u;
          const short2v cc = (__builtin_bit_cast(short2v, w[i]) + ev) >> 1;
          S = __builtin_amdgcn_sdot2(cc, one, S, false);
          const short2v a = __builtin_elementwise_min(__builtin_elementwise_max(cc, lo), hi);
          bv[i] = cc - a;
          ball |= __builtin_bit_cast(uint32_t, bv[i]);
          ab[i] = __builtin_bit_cast(uint32_t, a);
        }
        if (ball != 0) {  // rare: ~1e-5 of the entries at 3 350 hashes, one in 10^3 at 6 000
#pragma unroll
          for (int i = 0; i < 8; ++i) {
            const int32_t b = bv[i >> 1][i & 1];
            if (b != 0) {
              if (b > 127 || b < -127) bad |= 4u;
              const uint32_t idx = atomicAdd(&s_n[wv], 1u);
              if (idx < I8_ROW_ENT_MAX) s_ent[wv][idx] = (d0 + i) | ((uint32_t)(uint8_t)(int8_t)b << 16);
            }
          }
        }
        // low bytes of the eight clamped values: bytes 0 and 2 of every pair
        *reinterpret_cast<uint2 *>(out_a + (size_t)row * ldk8 + d0) =
            make_uint2(__builtin_amdgcn_perm(ab[1], ab[0], 0x06040200u), __builtin_amdgcn_perm(ab[3], ab[2], 0x06040200u));
        continue;
      }
      // the ragged end of a row (hv_d % 8 != 0), a misaligned matrix, the zero padding up to the K-step: value by value
      int32_t x[8];
#pragma unroll
      for (int i = 0; i < 8; ++i) x[i] = d0 + i < hv_d ? (int32_t)src[d0 + i] : -e;  // padding: c = 0
      uint32_t pk[2] = {0, 0};
#pragma unroll
      for (int i = 0; i < 8; ++i) {
        par |= d0 + i < hv_d ? (uint32_t)((x[i] ^ x0) & 1) : 0u;
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

}  // namespace
