// hg_encode_kernels.hip -- set semantics (sort + unique) and hypervector encode on gfx950.
//
// sort_unique : the HashSet<u64> of src/sketch.rs:93 / src/sketch_cuda.rs:158-163, as an
//               ascending duplicate-free list per genome (bitonic sort in LDS, one workgroup
//               per genome; a global-memory variant covers genomes whose hit count exceeds
//               the LDS budget).
// encode      : hd::encode_hash_hd{,_avx2} + dist::compute_hv_l2_norm
//               (src/hd.rs:14-112, src/dist.rs:132-137):
//                   hv[d] = 2 * #{h : bit_d(WyRng_h) = 1} - n      (i16 wrapping)
//               The WyRng stream is random-access (state_i = h + (i+1)*INC), so lane i of a
//               wave produces word i of every hash directly; the 64 bit-columns of that word
//               are counted with bit-sliced carry-save adders (a Harley-Seal tree over 16
//               hashes, ~6 logic ops per word instead of 128 per-bit adds), expanded once
//               per genome into LDS counters, and written out in the scalar or the AVX2
//               dimension order.
#include "hg_internal.h"

namespace {

constexpr int SORT_WG = 512;
constexpr uint32_t SORT_LDS_MAX_KEYS = HG_SORT_LDS_MAX_KEYS;  // 128 KiB of the 160 KiB LDS

// one compare-exchange pass of the bitonic network over a[0..n2), n2 a power of two
template <class Ptr>
__device__ __forceinline__ void bitonic_sort(Ptr a, uint32_t n2, uint32_t tid, uint32_t nthr) {
  for (uint32_t k = 2; k <= n2; k <<= 1) {
    for (uint32_t j = k >> 1; j > 0; j >>= 1) {
      for (uint32_t t = tid; t < (n2 >> 1); t += nthr) {
        // t-th pair of this stage
        uint32_t lo = ((t & ~(j - 1)) << 1) | (t & (j - 1));
        uint32_t hi = lo | j;
        bool up = (lo & k) == 0;
        uint64_t x = a[lo], y = a[hi];
        if ((x > y) == up) {
          a[lo] = y;
          a[hi] = x;
        }
      }
      __syncthreads();
    }
  }
}

__device__ __forceinline__ uint32_t next_pow2(uint32_t v) {
  uint32_t p = 1;
  while (p < v) p <<= 1;
  return p;
}

// block-wide exclusive scan of one uint per thread (SORT_WG threads); returns the
// exclusive prefix, *total = block sum.  scratch: SORT_WG/64 + 1 uints of LDS.
__device__ __forceinline__ uint32_t block_excl_scan(uint32_t v, uint32_t *scratch, uint32_t *total) {
  const uint32_t lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  uint32_t incl = v;
#pragma unroll
  for (int o = 1; o < 64; o <<= 1) {
    uint32_t n = __shfl_up(incl, o);
    if (lane >= (uint32_t)o) incl += n;
  }
  if (lane == 63) scratch[wave] = incl;
  __syncthreads();
  if (threadIdx.x == 0) {
    uint32_t run = 0;
    for (uint32_t w = 0; w < blockDim.x / 64; ++w) {
      uint32_t s = scratch[w];
      scratch[w] = run;
      run += s;
    }
    scratch[blockDim.x / 64] = run;
  }
  __syncthreads();
  uint32_t r = scratch[wave] + incl - v;
  *total = scratch[blockDim.x / 64];
  __syncthreads();
  return r;
}

// One workgroup per genome.  USE_LDS: keys are staged in dynamic LDS; otherwise the sort runs
// in place in the genome's hit region (which must have next_pow2(count) slots).
template <bool USE_LDS>
__global__ __launch_bounds__(SORT_WG) void sort_unique_kernel(
    const hg_genome_meta *__restrict__ meta, uint64_t *__restrict__ hits,
    const uint32_t *__restrict__ cnt, uint32_t *__restrict__ ndistinct, uint32_t lds_keys) {
  extern __shared__ __attribute__((aligned(16))) uint64_t s_keys[];
  __shared__ uint32_t s_scan[SORT_WG / 64 + 1];
  const uint32_t g = blockIdx.x;
  const hg_genome_meta gm = meta[g];
  uint32_t n = cnt[g];
  if (n > gm.hit_cap) n = gm.hit_cap;  // overflow is reported by the host from cnt[]
  uint64_t *region = hits + gm.hit_off;
  const uint32_t n2 = next_pow2(n);
  const bool in_lds = n2 <= lds_keys;
  if (USE_LDS != in_lds) return;  // the other instantiation handles this genome
  const uint32_t tid = threadIdx.x;

  if (n <= 1) {
    if (tid == 0) ndistinct[g] = n;
    return;
  }
  if (in_lds) {
    for (uint32_t i = tid; i < n2; i += SORT_WG) s_keys[i] = (i < n) ? region[i] : ~0ull;
    __syncthreads();
    bitonic_sort(s_keys, n2, tid, SORT_WG);
  } else {
    for (uint32_t i = n + tid; i < n2; i += SORT_WG) region[i] = ~0ull;  // hashes are < threshold < ~0
    __syncthreads();
    bitonic_sort(region, n2, tid, SORT_WG);
  }
  // unique: element i survives iff it differs from its predecessor; chunked scan + scatter.
  // In-place scatter is safe chunk by chunk only through a staging read, so read the whole
  // chunk into registers first, sync, then write (destination index <= source index).
  uint32_t base = 0;
  for (uint32_t c0 = 0; c0 < n; c0 += SORT_WG) {
    const uint32_t i = c0 + tid;
    uint64_t v = 0;
    uint32_t keep = 0;
    if (i < n) {
      v = in_lds ? s_keys[i] : region[i];
      uint64_t prev = (i == 0) ? ~v : (in_lds ? s_keys[i - 1] : region[i - 1]);
      keep = (v != prev) ? 1u : 0u;
    }
    __syncthreads();  // all reads of this chunk (incl. the i-1 neighbour) done
    uint32_t total;
    uint32_t pos = block_excl_scan(keep, s_scan, &total);
    if (keep) region[base + pos] = v;  // base+pos <= i: never overtakes an unread element
    base += total;
    __syncthreads();
  }
  if (tid == 0) ndistinct[g] = base;
}

// ---- encode -------------------------------------------------------------------------------
constexpr int ENC_WG = 512;
constexpr int ENC_WAVES = ENC_WG / 64;
constexpr uint64_t WY_INC = 0xa0761d6478bd642full;
constexpr uint64_t WY_XOR = 0xe7037ed1a0b428dbull;

__device__ __forceinline__ uint64_t wy_word(uint64_t hash, uint64_t off) {
  // word (i) of WyRng seeded with `hash`: state = hash + (i+1)*INC, off = (i+1)*INC
  uint64_t s = hash + off;
  unsigned __int128 m = (unsigned __int128)(s ^ WY_XOR) * s;
  return (uint64_t)(m >> 64) ^ (uint64_t)m;
}

// carry-save adder on 64 independent bit columns
__device__ __forceinline__ void csa(uint64_t &hi, uint64_t &lo, uint64_t a, uint64_t b, uint64_t c) {
  uint64_t u = a ^ b;
  hi = (a & b) | (u & c);
  lo = u ^ c;
}

constexpr int HI_PLANES = 10;  // counts up to 15 + 16*1023 per flush window

// One workgroup per genome.  hv_d/64 words are spread over lanes; when hv_d/64 < 64*ENC_WAVES
// several waves share a word and split the hashes.
__global__ __launch_bounds__(ENC_WG) void encode_kernel(
    const hg_genome_meta *__restrict__ meta, const uint64_t *__restrict__ hits,
    const uint32_t *__restrict__ ndistinct, uint32_t hv_d, uint32_t layout,
    int16_t *__restrict__ hv_out, int32_t *__restrict__ norm2_out) {
  extern __shared__ __attribute__((aligned(16))) uint32_t s_cnt[];  // [64][n_words + 1]
  __shared__ int32_t s_red[ENC_WAVES];
  const uint32_t g = blockIdx.x;
  const hg_genome_meta gm = meta[g];
  const uint32_t n = ndistinct[g];
  const uint64_t *__restrict__ hs = hits + gm.hit_off;
  const uint32_t n_words = hv_d / 64;
  const uint32_t stride = n_words + 1;  // +1: conflict-free column writes and row reads
  const uint32_t tid = threadIdx.x, lane = tid & 63;
  const uint32_t wave = __builtin_amdgcn_readfirstlane(tid >> 6);  // wave-uniform => scalar hash loads

  for (uint32_t i = tid; i < 64 * stride; i += ENC_WG) s_cnt[i] = 0;
  __syncthreads();

  // word groups of 64 words; (group, hash-slice) pairs are dealt round-robin to the waves
  const uint32_t n_groups = (n_words + 63) / 64;
  const uint32_t slices = (n_groups == 0 || n_groups >= (uint32_t)ENC_WAVES) ? 1u : (uint32_t)ENC_WAVES / n_groups;
  for (uint32_t job = wave; job < n_groups * slices; job += ENC_WAVES) {
    const uint32_t grp = job / slices, slice = job % slices;
    const uint32_t w = grp * 64 + lane;  // this lane's word index
    const bool w_ok = w < n_words;
    const uint64_t off = (uint64_t)(w + 1) * WY_INC;
    // hashes of this slice: blocks of 16, block b belongs to slice (b % slices)
    uint64_t ones = 0, twos = 0, fours = 0, eights = 0;
    uint64_t hp[HI_PLANES];
#pragma unroll
    for (int p = 0; p < HI_PLANES; ++p) hp[p] = 0;
    uint32_t blocks_in_window = 0;

    auto flush = [&]() {
      // expand the bit-sliced counters of this lane's word into the LDS counters
      if (w_ok) {
#pragma unroll 4
        for (uint32_t j = 0; j < 64; ++j) {
          uint32_t c = (uint32_t)((ones >> j) & 1) | ((uint32_t)((twos >> j) & 1) << 1) |
                       ((uint32_t)((fours >> j) & 1) << 2) | ((uint32_t)((eights >> j) & 1) << 3);
#pragma unroll
          for (int p = 0; p < HI_PLANES; ++p) c |= (uint32_t)((hp[p] >> j) & 1) << (4 + p);
          if (c) atomicAdd(&s_cnt[j * stride + w], c);
        }
      }
      ones = twos = fours = eights = 0;
#pragma unroll
      for (int p = 0; p < HI_PLANES; ++p) hp[p] = 0;
      blocks_in_window = 0;
    };

    for (uint32_t b0 = slice * 16; b0 < n; b0 += slices * 16) {
      uint64_t x[16];
#pragma unroll
      for (int t = 0; t < 16; ++t) {
        const uint32_t idx = b0 + t;
        x[t] = (idx < n) ? wy_word(hs[idx], off) : 0ull;  // hs[idx] is wave-uniform
      }
      // Harley-Seal: 16 inputs -> ones/twos/fours/eights + one carry into the 16s planes
      uint64_t twosA, twosB, foursA, foursB, eightsA, eightsB, sixteens;
      csa(twosA, ones, ones, x[0], x[1]);
      csa(twosB, ones, ones, x[2], x[3]);
      csa(foursA, twos, twos, twosA, twosB);
      csa(twosA, ones, ones, x[4], x[5]);
      csa(twosB, ones, ones, x[6], x[7]);
      csa(foursB, twos, twos, twosA, twosB);
      csa(eightsA, fours, fours, foursA, foursB);
      csa(twosA, ones, ones, x[8], x[9]);
      csa(twosB, ones, ones, x[10], x[11]);
      csa(foursA, twos, twos, twosA, twosB);
      csa(twosA, ones, ones, x[12], x[13]);
      csa(twosB, ones, ones, x[14], x[15]);
      csa(foursB, twos, twos, twosA, twosB);
      csa(eightsB, fours, fours, foursA, foursB);
      csa(sixteens, eights, eights, eightsA, eightsB);
      uint64_t carry = sixteens;  // ripple into the high planes
#pragma unroll
      for (int p = 0; p < HI_PLANES; ++p) {
        uint64_t t = hp[p] & carry;
        hp[p] ^= carry;
        carry = t;
      }
      if (++blocks_in_window == (1u << HI_PLANES) - 1) flush();
    }
    flush();
  }
  __syncthreads();

  // hv[d] = 2*count - n (i16 wrapping), laid out per `layout`; norm2 = sum hv^2 (i32 wrapping)
  uint32_t acc = 0;
  int16_t *__restrict__ out = hv_out + (size_t)g * hv_d;
  const uint32_t d_full = n_words * 64;
  for (uint32_t d = tid; d < hv_d; d += ENC_WG) {
    uint32_t c = 0;
    if (d < d_full) {
      const uint32_t w = d >> 6, pos = d & 63;
      // scalar: bit j -> pos j.  avx2: bit j -> pos 4*(j%16) + j/16, i.e. j = 16*(pos%4) + pos/4
      const uint32_t j = (layout == HG_LAYOUT_AVX2) ? (16 * (pos & 3) + (pos >> 2)) : pos;
      c = s_cnt[j * stride + w];
    }
    const int16_t v = (int16_t)(uint16_t)(2u * c - n);
    out[d] = v;
    acc += (uint32_t)((int32_t)v * (int32_t)v);
  }
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) acc += __shfl_down(acc, o);
  if (lane == 0) s_red[wave] = (int32_t)acc;
  __syncthreads();
  if (tid == 0) {
    uint32_t s = 0;
    for (int w = 0; w < ENC_WAVES; ++w) s += (uint32_t)s_red[w];
    norm2_out[g] = (int32_t)s;
  }
}

}  // namespace

hipError_t hg_launch_sort_unique(hipStream_t st, const hg_genome_meta *d_meta, uint32_t n_genomes,
                                 uint64_t *d_hits, const uint32_t *d_cnt, uint32_t *d_ndistinct,
                                 uint32_t max_cap) {
  if (n_genomes == 0) return hipSuccess;
  uint32_t keys = 1;
  while (keys < max_cap) keys <<= 1;
  if (keys > SORT_LDS_MAX_KEYS) keys = SORT_LDS_MAX_KEYS;
  const size_t lds = (size_t)keys * sizeof(uint64_t);
  static bool attr_set = false;
  if (!attr_set) {
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void *>(&sort_unique_kernel<true>),
                                       hipFuncAttributeMaxDynamicSharedMemorySize,
                                       SORT_LDS_MAX_KEYS * sizeof(uint64_t));
    if (e != hipSuccess) return e;
    attr_set = true;
  }
  hipLaunchKernelGGL((sort_unique_kernel<true>), dim3(n_genomes), dim3(SORT_WG), lds, st, d_meta,
                     d_hits, d_cnt, d_ndistinct, keys);
  hipError_t e = hipGetLastError();
  if (e != hipSuccess) return e;
  if (max_cap > SORT_LDS_MAX_KEYS) {  // some genome may need the in-place global sort
    hipLaunchKernelGGL((sort_unique_kernel<false>), dim3(n_genomes), dim3(SORT_WG), 0, st, d_meta,
                       d_hits, d_cnt, d_ndistinct, keys);
    e = hipGetLastError();
  }
  return e;
}

hipError_t hg_launch_encode(hipStream_t st, const hg_genome_meta *d_meta, uint32_t n_genomes,
                            const uint64_t *d_hits, const uint32_t *d_ndistinct, uint32_t hv_d,
                            uint32_t layout, int16_t *d_hv, int32_t *d_norm2) {
  if (n_genomes == 0) return hipSuccess;
  const size_t lds = (size_t)64 * (hv_d / 64 + 1) * sizeof(uint32_t);
  if (lds > 150 * 1024) return hipErrorInvalidValue;  // hv_d up to ~38k
  static bool attr_set = false;
  if (!attr_set) {
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void *>(&encode_kernel),
                                       hipFuncAttributeMaxDynamicSharedMemorySize, 150 * 1024);
    if (e != hipSuccess) return e;
    attr_set = true;
  }
  hipLaunchKernelGGL(encode_kernel, dim3(n_genomes), dim3(ENC_WG), lds, st, d_meta, d_hits,
                     d_ndistinct, hv_d, layout, d_hv, d_norm2);
  return hipGetLastError();
}
