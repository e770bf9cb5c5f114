// hg_hits.hip -- device-side post-processing of ANI hit lists:
//   * hg_sort_ani_hits_dev : the file order of dump_ani_file (src/utils.rs:262-269) -- stable ascending sort by
//                            ANI over the pair enumeration (row-major, src/dist.rs:251-265), then reversed
//                            = descending ANI, ties in REVERSE enumeration order;
//   * hg_topk_per_query_dev: the body of the reference's empty `search` subcommand (src/main.rs:22-24): per
//                            query the k best references, descending ANI (ties: lower reference index first).
// Both are LSD passes of a stable radix sort over an index permutation (keys are re-gathered between passes,
// the 12-byte hits move once at the end).  The sort is this file's own (radix_hist / radix_scan / radix_scatter_kernel:
// 8-bit digits, one histogram + one row scan + one stable scatter per digit, only the digits the keys can have: 1.29 M
// hits are ordered in 0.37 ms).  Until round 4 the passes were rocPRIM's device primitive (~1 ms, and ~10 ms for the first
// call of a process -- which is the only call `hyper-gen dist` makes).
#include <cstring>

#include "hg_internal.h"

namespace {

// key extraction for one LSD pass: 0 = enumeration key ref*Q+qry (u64), 1 = ANI bits (u32; ANI >= 0 so the
// IEEE bit pattern is monotone), 2 = ref index, 3 = query index
template <int WHAT>
__global__ __launch_bounds__(256) void gather_keys_kernel(const hg_ani_hit *__restrict__ hits, const uint32_t *__restrict__ perm,
                                                          uint32_t n, uint64_t Q, uint64_t *__restrict__ k64,
                                                          uint32_t *__restrict__ k32, uint32_t *__restrict__ iota) {
  const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  const uint32_t src = perm ? perm[i] : i;
  if (iota) iota[i] = i;
  const hg_ani_hit h = hits[src];
  if (WHAT == 0) k64[i] = (uint64_t)h.ref_idx * Q + h.qry_idx;
  else if (WHAT == 1) k32[i] = __float_as_uint(h.ani);
  else if (WHAT == 2) k32[i] = h.ref_idx;
  else k32[i] = h.qry_idx;
}

__global__ __launch_bounds__(256) void permute_hits_kernel(const hg_ani_hit *__restrict__ in, const uint32_t *__restrict__ perm,
                                                           uint32_t n, hg_ani_hit *__restrict__ out) {
  const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) out[i] = in[perm[i]];
}

// sorted by (qry asc, ani desc, ref asc): element i belongs to its query's top k iff the element k places
// earlier belongs to another query; its rank is i - (first element of the query), found by binary search
__global__ __launch_bounds__(256) void topk_kernel(const hg_ani_hit *__restrict__ sorted, uint32_t n, uint32_t Q, uint32_t k,
                                                   hg_ani_hit *__restrict__ out, uint32_t *__restrict__ counts) {
  const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  const hg_ani_hit h = sorted[i];
  if (h.qry_idx >= Q) return;
  if (i >= k && sorted[i - k].qry_idx == h.qry_idx) return;  // rank >= k
  uint32_t lo = 0, hi = i;  // first index whose qry_idx == h.qry_idx
  while (lo < hi) {
    const uint32_t mid = (lo + hi) / 2;
    if (sorted[mid].qry_idx < h.qry_idx) lo = mid + 1;
    else hi = mid;
  }
  const uint32_t rank = i - lo;
  out[(size_t)h.qry_idx * k + rank] = h;
  atomicMax(&counts[h.qry_idx], rank + 1);
}

__global__ __launch_bounds__(256) void fill_empty_kernel(hg_ani_hit *out, size_t n) {
  const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) out[i] = hg_ani_hit{0xFFFFFFFFu, 0xFFFFFFFFu, 0.f};
}

// ---- stable LSD radix sort of (key, value) pairs, one 8-bit digit per pass ----------------------------------------------
// A block owns a tile of RS_TILE consecutive elements; wave w of the block owns the tile's w-th quarter and walks it in
// rounds of 64 consecutive elements, so "earlier in the input" = (earlier wave, earlier round, lower lane).
//   radix_hist_kernel    counts[d * n_blocks + b] = elements of tile b whose digit is d
//   radix_scan_kernel    row d: exclusive prefix over the tiles (in place), totals[d] = the row's sum
//   radix_scatter_kernel element -> base(d) + prefix[d][b] + (same digit in earlier waves of the tile) + (same digit
//                        earlier in this wave's quarter); the last term comes from ballots over the digit's bits, so
//                        equal keys keep their input order: the passes compose like any LSD sort.
// DESC: the digit is taken from ~key, which turns the ascending pass into a stable descending one.
constexpr uint32_t RS_ITEMS = 16, RS_TILE = 256 * RS_ITEMS, RS_QUARTER = 64 * RS_ITEMS;

template <class K, bool DESC>
__device__ __forceinline__ uint32_t radix_digit(K key, uint32_t shift, uint32_t mask) {
  return (uint32_t)((DESC ? ~key : key) >> shift) & mask;
}

template <class K, bool DESC>
__global__ __launch_bounds__(256) void radix_hist_kernel(const K *__restrict__ keys, uint32_t n, uint32_t shift, uint32_t mask,
                                                         uint32_t n_blocks, uint32_t *__restrict__ counts) {
  __shared__ uint32_t s_h[256];
  s_h[threadIdx.x] = 0;
  __syncthreads();
  const uint32_t base = blockIdx.x * RS_TILE;
#pragma unroll 4
  for (uint32_t i = 0; i < RS_ITEMS; ++i) {
    const uint32_t idx = base + i * 256 + threadIdx.x;
    if (idx < n) atomicAdd(&s_h[radix_digit<K, DESC>(keys[idx], shift, mask)], 1u);
  }
  __syncthreads();
  counts[(size_t)threadIdx.x * n_blocks + blockIdx.x] = s_h[threadIdx.x];
}

// block-wide exclusive scan of one value per thread (256 threads); returns the block total through *total
__device__ __forceinline__ uint32_t block_excl_scan_256(uint32_t v, uint32_t *s_wave /* 4 words */, uint32_t *total) {
  const uint32_t lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  uint32_t inc = v;
#pragma unroll
  for (int o = 1; o < 64; o <<= 1) {
    const uint32_t up = __shfl_up(inc, o);
    if (lane >= (uint32_t)o) inc += up;
  }
  if (lane == 63) s_wave[wave] = inc;
  __syncthreads();
  uint32_t before = 0, all = 0;
#pragma unroll
  for (uint32_t w = 0; w < 4; ++w) {
    const uint32_t t = s_wave[w];
    before += w < wave ? t : 0u, all += t;
  }
  __syncthreads();  // (s_wave may be reused by the caller's next round)
  *total = all;
  return before + inc - v;
}

__global__ __launch_bounds__(256) void radix_scan_kernel(uint32_t *__restrict__ counts, uint32_t n_blocks, uint32_t *__restrict__ totals) {
  __shared__ uint32_t s_wave[4];
  uint32_t *row = counts + (size_t)blockIdx.x * n_blocks;
  uint32_t running = 0;
  for (uint32_t c = 0; c < n_blocks; c += 256) {
    const uint32_t i = c + threadIdx.x, v = i < n_blocks ? row[i] : 0u;
    uint32_t tot;
    const uint32_t ex = block_excl_scan_256(v, s_wave, &tot);
    if (i < n_blocks) row[i] = running + ex;
    running += tot;
  }
  if (threadIdx.x == 0) totals[blockIdx.x] = running;
}

template <class K, bool DESC>
__global__ __launch_bounds__(256) void radix_scatter_kernel(const K *__restrict__ keys, const uint32_t *__restrict__ vals, uint32_t n,
                                                            uint32_t shift, uint32_t mask, uint32_t n_blocks,
                                                            const uint32_t *__restrict__ counts, const uint32_t *__restrict__ totals,
                                                            K *__restrict__ keys_out, uint32_t *__restrict__ vals_out) {
  __shared__ uint32_t s_cnt[4][256];  // per wave and digit: elements seen so far, then the waves' exclusive prefix
  __shared__ uint32_t s_base[256];    // where the tile's elements of digit d start in the output
  __shared__ uint32_t s_wave[4];
  const uint32_t lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  {
    uint32_t tot;
    const uint32_t dbase = block_excl_scan_256(totals[threadIdx.x], s_wave, &tot);
    s_base[threadIdx.x] = dbase + counts[(size_t)threadIdx.x * n_blocks + blockIdx.x];
#pragma unroll
    for (uint32_t w = 0; w < 4; ++w) s_cnt[w][threadIdx.x] = 0;
  }
  __syncthreads();
  const uint32_t q0 = blockIdx.x * RS_TILE + wave * RS_QUARTER;
  K key[RS_ITEMS];
  uint32_t rank[RS_ITEMS];
  const unsigned long long lt = lane ? (~0ull >> (64 - lane)) : 0ull;
#pragma unroll
  for (uint32_t r = 0; r < RS_ITEMS; ++r) {
    const uint32_t idx = q0 + r * 64 + lane;
    const bool valid = idx < n;
    key[r] = valid ? keys[idx] : (K)0;
    const uint32_t d = radix_digit<K, DESC>(key[r], shift, mask);
    unsigned long long peers = __ballot(valid);
#pragma unroll
    for (uint32_t b = 0; b < 8; ++b) {
      const unsigned long long m = __ballot((d >> b) & 1u);
      peers &= ((d >> b) & 1u) ? m : ~m;
    }
    // (a wave's LDS operations execute in program order: every lane of a digit's group reads the count before the
    // group's first lane moves it on.  The lanes talk to each other through these words, so the accesses are relaxed
    // atomics -- plain ds_read / ds_write, but never cached in a register from one round to the next)
    const uint32_t seen = __hip_atomic_load(&s_cnt[wave][d], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
    rank[r] = seen + (uint32_t)__popcll(peers & lt);
    __builtin_amdgcn_wave_barrier();
    if (valid && (peers & lt) == 0ull)
      __hip_atomic_store(&s_cnt[wave][d], seen + (uint32_t)__popcll(peers), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
    __builtin_amdgcn_wave_barrier();
  }
  __syncthreads();
  {  // digit threadIdx.x: the waves' counts -> exclusive prefix over the waves
    uint32_t run = 0;
#pragma unroll
    for (uint32_t w = 0; w < 4; ++w) {
      const uint32_t c = s_cnt[w][threadIdx.x];
      s_cnt[w][threadIdx.x] = run;
      run += c;
    }
  }
  __syncthreads();
#pragma unroll
  for (uint32_t r = 0; r < RS_ITEMS; ++r) {
    const uint32_t idx = q0 + r * 64 + lane;
    if (idx < n) {
      const uint32_t d = radix_digit<K, DESC>(key[r], shift, mask), pos = s_base[d] + s_cnt[wave][d] + rank[r];
      keys_out[pos] = key[r];
      vals_out[pos] = vals[idx];
    }
  }
}

// keys[begin_bit, end_bit) decide; returns through *flipped whether the result is in the `b` arrays (odd number of passes)
template <class K, bool DESC>
hipError_t radix_sort_pairs(hipStream_t st, uint32_t *d_counts, K *ka, K *kb, uint32_t *va, uint32_t *vb, uint32_t n,
                            uint32_t begin_bit, uint32_t end_bit, bool *flipped) {
  const uint32_t n_blocks = (n + RS_TILE - 1) / RS_TILE;
  uint32_t *d_totals = d_counts + (size_t)256 * n_blocks;
  bool flip = false;
  for (uint32_t shift = begin_bit; shift < end_bit; shift += 8) {
    const uint32_t bits = end_bit - shift < 8 ? end_bit - shift : 8, mask = (1u << bits) - 1u;
    const K *kin = flip ? kb : ka;
    const uint32_t *vin = flip ? vb : va;
    hipLaunchKernelGGL((radix_hist_kernel<K, DESC>), dim3(n_blocks), dim3(256), 0, st, kin, n, shift, mask, n_blocks, d_counts);
    hipLaunchKernelGGL(radix_scan_kernel, dim3(256), dim3(256), 0, st, d_counts, n_blocks, d_totals);
    hipLaunchKernelGGL((radix_scatter_kernel<K, DESC>), dim3(n_blocks), dim3(256), 0, st, kin, vin, n, shift, mask, n_blocks,
                       d_counts, d_totals, flip ? ka : kb, flip ? va : vb);
    const hipError_t e = hipGetLastError();
    if (e != hipSuccess) return e;
    flip = !flip;
  }
  *flipped = flip;
  return hipSuccess;
}
inline uint32_t bits_for(uint64_t max_value) {  // digits above the highest set bit of the largest possible key are all equal
  uint32_t b = 1;
  while (b < 64 && (max_value >> b)) ++b;
  return b;
}

struct SortWs {
  uint64_t *k64a, *k64b;
  uint32_t *k32a, *k32b, *va, *vb;
  hg_ani_hit *tmp_hits;
  void *tmp;
  size_t tmp_bytes;
};

hg_status sort_workspace(hg_ctx *c, uint32_t n, SortWs &w) {
  const size_t n_blocks = ((size_t)n + RS_TILE - 1) / RS_TILE;
  const size_t tb = ((256 * n_blocks + 256) * sizeof(uint32_t) + 255) & ~(size_t)255;  // digit counts per tile + digit totals
  const size_t a8 = ((size_t)n * 8 + 255) & ~(size_t)255, a4 = ((size_t)n * 4 + 255) & ~(size_t)255;
  const size_t ah = ((size_t)n * sizeof(hg_ani_hit) + 255) & ~(size_t)255;
  hg_status s = hg_ensure(c, c->w_sorthits, 2 * a8 + 4 * a4 + ah + tb + 256);
  if (s != HG_OK) return s;
  uint8_t *p = static_cast<uint8_t *>(c->w_sorthits.p);
  w.k64a = reinterpret_cast<uint64_t *>(p), p += a8;
  w.k64b = reinterpret_cast<uint64_t *>(p), p += a8;
  w.k32a = reinterpret_cast<uint32_t *>(p), p += a4;
  w.k32b = reinterpret_cast<uint32_t *>(p), p += a4;
  w.va = reinterpret_cast<uint32_t *>(p), p += a4;
  w.vb = reinterpret_cast<uint32_t *>(p), p += a4;
  w.tmp_hits = reinterpret_cast<hg_ani_hit *>(p), p += ah;
  w.tmp = p, w.tmp_bytes = tb;
  return HG_OK;
}

}  // namespace

extern "C" hg_status hg_sort_ani_hits_dev(hg_ctx *c, hg_ani_hit *d_hits, size_t n, size_t Q) {
  if (!c) return HG_ERR_INVALID;
  if (n < 2) return HG_OK;
  if (!d_hits) return hg_fail(c, HG_ERR_INVALID, "NULL hit list");
  // (the sort's kernels index elements in 32 bits, a tile of RS_TILE at a time: the last tile's indices must not wrap)
  if (n > 0xFFFFFFFFull - RS_TILE) return hg_fail(c, HG_ERR_UNSUPPORTED, "hit list too long for the device sort");
  HG_ENTER(c);
  const uint32_t m = (uint32_t)n, grid = (m + 255) / 256;
  SortWs w;
  hg_status s = sort_workspace(c, m, w);
  if (s != HG_OK) return s;
  // pass 1 (least significant): enumeration key ref * Q + qry, descending -- only the bits it can have
  auto *cnt = static_cast<uint32_t *>(w.tmp);
  hipLaunchKernelGGL(gather_keys_kernel<0>, dim3(grid), dim3(256), 0, c->stream, d_hits, (const uint32_t *)nullptr, m,
                     (uint64_t)Q, w.k64a, (uint32_t *)nullptr, w.va);
  HG_HIP(c, hipGetLastError());
  bool f1 = false, f2 = false;
  // ref_idx < 2^32, so the key is below 2^32 * Q: bits_for(Q << 32) digits at most; a caller's list usually needs far fewer,
  // but the highest reference index is not known on the host
  HG_HIP(c, (radix_sort_pairs<uint64_t, true>(c->stream, cnt, w.k64a, w.k64b, w.va, w.vb, m, 0,
                                              std::min<uint32_t>(64, 32 + bits_for((uint64_t)(Q ? Q - 1 : 0))), &f1)));
  uint32_t *perm1 = f1 ? w.vb : w.va, *other = f1 ? w.va : w.vb;
  // pass 2 (most significant): ANI, descending; the sort is stable, so ties keep pass 1's order
  hipLaunchKernelGGL(gather_keys_kernel<1>, dim3(grid), dim3(256), 0, c->stream, d_hits, perm1, m, (uint64_t)Q,
                     (uint64_t *)nullptr, w.k32a, (uint32_t *)nullptr);
  HG_HIP(c, hipGetLastError());
  HG_HIP(c, (radix_sort_pairs<uint32_t, true>(c->stream, cnt, w.k32a, w.k32b, perm1, other, m, 0, 32, &f2)));
  const uint32_t *perm2 = f2 ? other : perm1;
  hipLaunchKernelGGL(permute_hits_kernel, dim3(grid), dim3(256), 0, c->stream, d_hits, perm2, m, w.tmp_hits);
  HG_HIP(c, hipGetLastError());
  HG_HIP(c, hipMemcpyAsync(d_hits, w.tmp_hits, (size_t)m * sizeof(hg_ani_hit), hipMemcpyDeviceToDevice, c->stream));
  return HG_OK;
}

extern "C" hg_status hg_sort_ani_hits_staged(hg_ctx *c, hg_ani_hit *hits, size_t n, size_t Q) {
  if (!c) return HG_ERR_INVALID;
  if (n < 2) return HG_OK;
  if (!hits) return hg_fail(c, HG_ERR_INVALID, "NULL hit list");
  HG_ENTER(c);
  hg_status s = hg_ensure(c, c->w_ani, n * sizeof(hg_ani_hit) + 64);
  if (s != HG_OK) return s;
  HG_HIP(c, hipMemcpyAsync(c->w_ani.p, hits, n * sizeof(hg_ani_hit), hipMemcpyHostToDevice, c->stream));
  if ((s = hg_sort_ani_hits_dev(c, static_cast<hg_ani_hit *>(c->w_ani.p), n, Q)) != HG_OK) return s;
  HG_HIP(c, hipMemcpyAsync(hits, c->w_ani.p, n * sizeof(hg_ani_hit), hipMemcpyDeviceToHost, c->stream));
  HG_HIP(c, hipStreamSynchronize(c->stream));
  return HG_OK;
}

extern "C" hg_status hg_topk_per_query_dev(hg_ctx *c, const hg_ani_hit *d_hits, size_t n, size_t Q, uint32_t k,
                                           hg_ani_hit *d_out, uint32_t *d_counts) {
  if (!c) return HG_ERR_INVALID;
  if (Q == 0 || k == 0) return HG_OK;
  if (!d_out || !d_counts || (n && !d_hits)) return hg_fail(c, HG_ERR_INVALID, "NULL argument");
  if (n > 0xFFFFFFFFull - RS_TILE || Q > 0xFFFFFFF0ull) return hg_fail(c, HG_ERR_UNSUPPORTED, "hit list too long for the device sort");
  HG_ENTER(c);
  const size_t slots = Q * (size_t)k;
  hipLaunchKernelGGL(fill_empty_kernel, dim3((unsigned)((slots + 255) / 256)), dim3(256), 0, c->stream, d_out, slots);
  HG_HIP(c, hipGetLastError());
  HG_HIP(c, hipMemsetAsync(d_counts, 0, Q * sizeof(uint32_t), c->stream));
  if (n == 0) return HG_OK;
  const uint32_t m = (uint32_t)n, grid = (m + 255) / 256;
  SortWs w;
  hg_status s = sort_workspace(c, m, w);
  if (s != HG_OK) return s;
  auto *cnt = static_cast<uint32_t *>(w.tmp);
  bool f = false;
  // LSD order: reference index ascending, then ANI descending, then query index ascending (all stable)
  hipLaunchKernelGGL(gather_keys_kernel<2>, dim3(grid), dim3(256), 0, c->stream, d_hits, (const uint32_t *)nullptr, m,
                     (uint64_t)Q, (uint64_t *)nullptr, w.k32a, w.va);
  HG_HIP(c, hipGetLastError());
  HG_HIP(c, (radix_sort_pairs<uint32_t, false>(c->stream, cnt, w.k32a, w.k32b, w.va, w.vb, m, 0, 32, &f)));
  uint32_t *perm = f ? w.vb : w.va, *other = f ? w.va : w.vb;
  hipLaunchKernelGGL(gather_keys_kernel<1>, dim3(grid), dim3(256), 0, c->stream, d_hits, perm, m, (uint64_t)Q,
                     (uint64_t *)nullptr, w.k32a, (uint32_t *)nullptr);
  HG_HIP(c, hipGetLastError());
  HG_HIP(c, (radix_sort_pairs<uint32_t, true>(c->stream, cnt, w.k32a, w.k32b, perm, other, m, 0, 32, &f)));
  if (f) std::swap(perm, other);
  hipLaunchKernelGGL(gather_keys_kernel<3>, dim3(grid), dim3(256), 0, c->stream, d_hits, perm, m, (uint64_t)Q,
                     (uint64_t *)nullptr, w.k32a, (uint32_t *)nullptr);
  HG_HIP(c, hipGetLastError());
  // (hits with qry_idx >= Q are dropped by topk_kernel; their keys may use all 32 bits)
  HG_HIP(c, (radix_sort_pairs<uint32_t, false>(c->stream, cnt, w.k32a, w.k32b, perm, other, m, 0, 32, &f)));
  if (f) std::swap(perm, other);
  hipLaunchKernelGGL(permute_hits_kernel, dim3(grid), dim3(256), 0, c->stream, d_hits, perm, m, w.tmp_hits);
  HG_HIP(c, hipGetLastError());
  hipLaunchKernelGGL(topk_kernel, dim3(grid), dim3(256), 0, c->stream, w.tmp_hits, m, (uint32_t)Q, k, d_out, d_counts);
  HG_HIP(c, hipGetLastError());
  return HG_OK;
}
