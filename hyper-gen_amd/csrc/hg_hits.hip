// hg_hits.hip -- device-side post-processing of ANI hit lists:
//   * hg_sort_ani_hits_dev : the file order of dump_ani_file (src/utils.rs:262-269) -- stable ascending sort by
//                            ANI over the pair enumeration (row-major, src/dist.rs:251-265), then reversed
//                            = descending ANI, ties in REVERSE enumeration order;
//   * hg_topk_per_query_dev: the body of the reference's empty `search` subcommand (src/main.rs:22-24): per
//                            query the k best references, descending ANI (ties: lower reference index first).
// Both are LSD passes of a stable radix sort over an index permutation (keys are re-gathered between passes,
// the 12-byte hits move once at the end).  The radix-sort passes are rocPRIM's device primitive (AMD's own
// header-only primitives library under /opt/rocm/include; not a compatibility layer) -- ordering a hit list is
// byte shuffling off the hot path; the hot kernels (hash, encode, GEMM, popcount) stay hand-written.
#include <cstring>

#include <rocprim/device/device_radix_sort.hpp>

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

struct SortWs {
  uint64_t *k64a, *k64b;
  uint32_t *k32a, *k32b, *va, *vb;
  hg_ani_hit *tmp_hits;
  void *tmp;
  size_t tmp_bytes;
};

hg_status sort_workspace(hg_ctx *c, uint32_t n, SortWs &w) {
  size_t t64 = 0, t32 = 0;
  HG_HIP(c, rocprim::radix_sort_pairs_desc(nullptr, t64, (uint64_t *)nullptr, (uint64_t *)nullptr, (uint32_t *)nullptr,
                                           (uint32_t *)nullptr, n, 0, 64, c->stream));
  HG_HIP(c, rocprim::radix_sort_pairs_desc(nullptr, t32, (uint32_t *)nullptr, (uint32_t *)nullptr, (uint32_t *)nullptr,
                                           (uint32_t *)nullptr, n, 0, 32, c->stream));
  const size_t tb = (std::max(t64, t32) + 255) & ~(size_t)255;
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
  if (n > 0xFFFFFFF0ull) return hg_fail(c, HG_ERR_UNSUPPORTED, "hit list too long for the device sort");
  HG_HIP(c, hipSetDevice(c->device));
  const uint32_t m = (uint32_t)n, grid = (m + 255) / 256;
  SortWs w;
  hg_status s = sort_workspace(c, m, w);
  if (s != HG_OK) return s;
  // pass 1 (least significant): enumeration key, descending
  hipLaunchKernelGGL(gather_keys_kernel<0>, dim3(grid), dim3(256), 0, c->stream, d_hits, (const uint32_t *)nullptr, m,
                     (uint64_t)Q, w.k64a, (uint32_t *)nullptr, w.va);
  HG_HIP(c, hipGetLastError());
  size_t tb = w.tmp_bytes;
  HG_HIP(c, rocprim::radix_sort_pairs_desc(w.tmp, tb, w.k64a, w.k64b, w.va, w.vb, m, 0, 64, c->stream));
  // pass 2 (most significant): ANI, descending; the radix sort is stable, so ties keep pass 1's order
  hipLaunchKernelGGL(gather_keys_kernel<1>, dim3(grid), dim3(256), 0, c->stream, d_hits, w.vb, m, (uint64_t)Q,
                     (uint64_t *)nullptr, w.k32a, (uint32_t *)nullptr);
  HG_HIP(c, hipGetLastError());
  tb = w.tmp_bytes;
  HG_HIP(c, rocprim::radix_sort_pairs_desc(w.tmp, tb, w.k32a, w.k32b, w.vb, w.va, m, 0, 32, c->stream));
  hipLaunchKernelGGL(permute_hits_kernel, dim3(grid), dim3(256), 0, c->stream, d_hits, w.va, m, w.tmp_hits);
  HG_HIP(c, hipGetLastError());
  HG_HIP(c, hipMemcpyAsync(d_hits, w.tmp_hits, (size_t)m * sizeof(hg_ani_hit), hipMemcpyDeviceToDevice, c->stream));
  return HG_OK;
}

extern "C" hg_status hg_sort_ani_hits_staged(hg_ctx *c, hg_ani_hit *hits, size_t n, size_t Q) {
  if (!c) return HG_ERR_INVALID;
  if (n < 2) return HG_OK;
  if (!hits) return hg_fail(c, HG_ERR_INVALID, "NULL hit list");
  HG_HIP(c, hipSetDevice(c->device));
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
  if (n > 0xFFFFFFF0ull || Q > 0xFFFFFFF0ull) return hg_fail(c, HG_ERR_UNSUPPORTED, "hit list too long for the device sort");
  HG_HIP(c, hipSetDevice(c->device));
  const size_t slots = Q * (size_t)k;
  hipLaunchKernelGGL(fill_empty_kernel, dim3((unsigned)((slots + 255) / 256)), dim3(256), 0, c->stream, d_out, slots);
  HG_HIP(c, hipGetLastError());
  HG_HIP(c, hipMemsetAsync(d_counts, 0, Q * sizeof(uint32_t), c->stream));
  if (n == 0) return HG_OK;
  const uint32_t m = (uint32_t)n, grid = (m + 255) / 256;
  SortWs w;
  hg_status s = sort_workspace(c, m, w);
  if (s != HG_OK) return s;
  size_t tb;
  // LSD order: reference index ascending, then ANI descending, then query index ascending (all stable)
  hipLaunchKernelGGL(gather_keys_kernel<2>, dim3(grid), dim3(256), 0, c->stream, d_hits, (const uint32_t *)nullptr, m,
                     (uint64_t)Q, (uint64_t *)nullptr, w.k32a, w.va);
  HG_HIP(c, hipGetLastError());
  tb = w.tmp_bytes;
  HG_HIP(c, rocprim::radix_sort_pairs(w.tmp, tb, w.k32a, w.k32b, w.va, w.vb, m, 0, 32, c->stream));
  hipLaunchKernelGGL(gather_keys_kernel<1>, dim3(grid), dim3(256), 0, c->stream, d_hits, w.vb, m, (uint64_t)Q,
                     (uint64_t *)nullptr, w.k32a, (uint32_t *)nullptr);
  HG_HIP(c, hipGetLastError());
  tb = w.tmp_bytes;
  HG_HIP(c, rocprim::radix_sort_pairs_desc(w.tmp, tb, w.k32a, w.k32b, w.vb, w.va, m, 0, 32, c->stream));
  hipLaunchKernelGGL(gather_keys_kernel<3>, dim3(grid), dim3(256), 0, c->stream, d_hits, w.va, m, (uint64_t)Q,
                     (uint64_t *)nullptr, w.k32a, (uint32_t *)nullptr);
  HG_HIP(c, hipGetLastError());
  tb = w.tmp_bytes;
  HG_HIP(c, rocprim::radix_sort_pairs(w.tmp, tb, w.k32a, w.k32b, w.va, w.vb, m, 0, 32, c->stream));
  hipLaunchKernelGGL(permute_hits_kernel, dim3(grid), dim3(256), 0, c->stream, d_hits, w.vb, m, w.tmp_hits);
  HG_HIP(c, hipGetLastError());
  hipLaunchKernelGGL(topk_kernel, dim3(grid), dim3(256), 0, c->stream, w.tmp_hits, m, (uint32_t)Q, k, d_out, d_counts);
  HG_HIP(c, hipGetLastError());
  return HG_OK;
}
