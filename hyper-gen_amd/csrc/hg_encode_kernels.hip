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
#include <atomic>

#include "hg_internal.h"

namespace {

constexpr int SORT_WG = 512;
constexpr uint32_t SORT_LDS_MAX_KEYS = HG_SORT_LDS_MAX_KEYS;  // 64 KiB of keys + 32 KiB of counters of the 160 KiB LDS

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
// Counting-sort fast path of the LDS sort (bucket_mul != 0, 512 <= n, lds_keys <= SORT_BUCKET_MAX_KEYS): sampled
// hashes are uniform below the threshold, so the monotone map b = floor(h * n2 / threshold) puts 0.8 keys into each of
// n2 buckets on average.  Count (one returning LDS atomic per key: its rank inside the bucket), scan, scatter, then
// every thread orders the few keys of its buckets by insertion: five passes over the keys instead of the bitonic
// network's 78.  A bucket with more than SORT_BUCKET_LIMIT keys (repeats: equal hashes share a bucket) sends the genome
// to the bitonic sort after all -- the keys are in LDS by then.
constexpr uint32_t SORT_BUCKET_MAX_KEYS = 8192, SORT_BUCKET_LIMIT = 16, SORT_KPT = SORT_BUCKET_MAX_KEYS / SORT_WG;
static_assert(SORT_BUCKET_MAX_KEYS == SORT_LDS_MAX_KEYS, "every set the one-workgroup sort takes can take its counting sort");
constexpr size_t SORT_LDS_BYTES_MAX = (size_t)SORT_LDS_MAX_KEYS * (sizeof(uint64_t) + sizeof(uint32_t));  // keys + counters

// Up to 64 keys ordered and de-duplicated by ONE wave in registers (lane = the calling lane, 0..63): a 64-lane bitonic
// network over shuffles, no LDS, no barrier.
__device__ __forceinline__ void sort_unique_wave(uint64_t *__restrict__ region, const uint32_t n, const uint32_t lane,
                                                 uint32_t *__restrict__ nd_out) {
  uint64_t key = lane < n ? region[lane] : ~0ull;  // hashes are < threshold < ~0
#pragma unroll
  for (uint32_t k = 2; k <= 64; k <<= 1)
#pragma unroll
    for (uint32_t j = k >> 1; j > 0; j >>= 1) {
      const uint64_t other = ((uint64_t)(uint32_t)__shfl_xor((int)(uint32_t)(key >> 32), (int)j) << 32) |
                             (uint32_t)__shfl_xor((int)(uint32_t)key, (int)j);
      const bool lower = (lane & j) == 0, asc = (lane & k) == 0;
      const bool take_min = lower == asc;
      key = (take_min == (other < key)) ? other : key;
    }
  const uint64_t prev = ((uint64_t)(uint32_t)__shfl_up((int)(uint32_t)(key >> 32), 1) << 32) | (uint32_t)__shfl_up((int)(uint32_t)key, 1);
  const bool keep = lane < n && (lane == 0 || key != prev);
  const unsigned long long kb = __ballot(keep);
  if (keep) region[__popcll(kb & ((1ull << lane) - 1ull))] = key;  // (every key was read before the first one is written)
  if (lane == 0) *nd_out = (uint32_t)__popcll(kb);
}

// The sort + unique of ONE genome by the calling workgroup (n = its stored raw hits).  Returns early -- whole waves, or
// the whole workgroup -- on the paths that need no barrier; a caller that loops over genomes puts a barrier between them.
template <bool USE_LDS>
__device__ __forceinline__ void sort_unique_one(const uint32_t g, const hg_genome_meta &gm, const uint32_t n,
                                                uint64_t *__restrict__ hits, uint32_t *__restrict__ ndistinct,
                                                const uint32_t lds_keys, const uint64_t bucket_mul) {
  extern __shared__ __attribute__((aligned(16))) uint64_t s_keys[];
  __shared__ uint32_t s_scan[SORT_WG / 64 + 1];
  uint64_t *region = hits + gm.hit_off;
  const uint32_t n2 = next_pow2(n);
  const bool in_lds = n2 <= lds_keys;
  if (USE_LDS != in_lds) return;  // larger sets: bucketed sort (hg_launch_sort_large) or the in-place variant
  const uint32_t tid = threadIdx.x;

  if (n <= 1) {
    if (tid == 0) ndistinct[g] = n;
    return;
  }
  if (USE_LDS && n <= 64) {
    // A handful of hashes (plasmids, viral genomes, the 50 kbp genomes of bench.py's `many_small` leg: 33 hashes each): one
    // wave orders them in registers -- a 64-lane bitonic network over shuffles, no LDS, no barrier -- while the other waves
    // leave.  (Through the workgroup-wide network with its barrier per pass, 100 000 such genomes took 0.96 ms; the k-mer
    // kernel of the same batch 9.3 ms.)
    if (tid >= 64) return;  // whole waves
    sort_unique_wave(region, n, tid, ndistinct + g);
    return;
  }
  if (in_lds) {
    bool sorted = false;  // workgroup-uniform
    if (USE_LDS && bucket_mul != 0 && n >= (uint32_t)SORT_WG && lds_keys <= SORT_BUCKET_MAX_KEYS) {
      __shared__ uint32_t s_over;
      uint32_t *s_bk = reinterpret_cast<uint32_t *>(s_keys + lds_keys);  // n2 bucket counters, then bucket starts
      const uint32_t shift = (uint32_t)(__builtin_ctz(lds_keys) - __builtin_ctz(n2)), per = n2 / SORT_WG;
      for (uint32_t i = tid; i < n2; i += SORT_WG) s_bk[i] = 0;
      if (tid == 0) s_over = 0;
      __syncthreads();
      uint64_t kk[SORT_KPT];
      uint32_t bb[SORT_KPT], rr[SORT_KPT];
#pragma unroll
      for (uint32_t u = 0; u < SORT_KPT; ++u) {
        const uint32_t i = tid + u * SORT_WG;
        if (i < n) {
          kk[u] = region[i];
          const uint32_t b = (uint32_t)__umul64hi(kk[u], bucket_mul) >> shift;
          bb[u] = b < n2 ? b : n2 - 1;
          rr[u] = atomicAdd(&s_bk[bb[u]], 1u);
        }
      }
      __syncthreads();
      {  // exclusive scan of the counters: thread t owns buckets [t * per, (t + 1) * per)
        uint32_t c[SORT_KPT], sum = 0, mx = 0;
#pragma unroll
        for (uint32_t q = 0; q < SORT_KPT; ++q)
          if (q < per) c[q] = s_bk[tid * per + q], sum += c[q], mx = c[q] > mx ? c[q] : mx;
        if (mx > SORT_BUCKET_LIMIT) s_over = 1u;  // (same value from every writer)
        uint32_t total;
        uint32_t run = block_excl_scan(sum, s_scan, &total);
#pragma unroll
        for (uint32_t q = 0; q < SORT_KPT; ++q)
          if (q < per) s_bk[tid * per + q] = run, run += c[q];
      }
      __syncthreads();
#pragma unroll
      for (uint32_t u = 0; u < SORT_KPT; ++u)
        if (tid + u * SORT_WG < n) s_keys[s_bk[bb[u]] + rr[u]] = kk[u];
      __syncthreads();
      sorted = s_over == 0u;
      if (sorted) {
        for (uint32_t q = 0; q < per; ++q) {  // order the keys inside each of this thread's buckets
          const uint32_t b = tid * per + q, lo = s_bk[b], hi = b + 1 < n2 ? s_bk[b + 1] : n;
          for (uint32_t i = lo + 1; i < hi; ++i) {
            const uint64_t v = s_keys[i];
            uint32_t j = i;
            while (j > lo && s_keys[j - 1] > v) s_keys[j] = s_keys[j - 1], --j;
            s_keys[j] = v;
          }
        }
      } else {
        for (uint32_t i = n + tid; i < n2; i += SORT_WG) s_keys[i] = ~0ull;  // the keys are all here: bitonic after all
      }
      __syncthreads();
      if (!sorted) bitonic_sort(s_keys, n2, tid, SORT_WG);
    } else {
      for (uint32_t i = tid; i < n2; i += SORT_WG) s_keys[i] = (i < n) ? region[i] : ~0ull;
      __syncthreads();
      bitonic_sort(s_keys, n2, tid, SORT_WG);
    }
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

// One workgroup per genome (of the todo list, if there is one).
template <bool USE_LDS>
__global__ __launch_bounds__(SORT_WG) void sort_unique_kernel(
    const hg_genome_meta *__restrict__ meta, uint64_t *__restrict__ hits,
    const uint32_t *__restrict__ cnt, uint32_t *__restrict__ ndistinct, uint32_t lds_keys,
    const uint32_t *__restrict__ todo, uint64_t bucket_mul, uint32_t *__restrict__ flags) {
  const uint32_t g = todo ? todo[blockIdx.x] : blockIdx.x;
  const hg_genome_meta gm = meta[g];
  uint32_t n = cnt[g];
  if (flags) {
    // the sync-free step: nobody on the host looks at cnt[] before the encoders run -- a genome this launch and its
    // hg_launch_sort_unique_rest cannot finish is marked for them and reported through the step's flag word
    const bool over = n > gm.hit_cap;
    if (over || n > SORT_LDS_MAX_KEYS) {
      if (threadIdx.x == 0) atomicOr(flags, over ? HG_STEP_OVERFLOW : HG_STEP_LARGE_SET), ndistinct[g] = HG_NHASH_PENDING;
      return;
    }
  }
  if (n > gm.hit_cap) n = gm.hit_cap;  // overflow is reported by the host from cnt[]
  sort_unique_one<USE_LDS>(g, gm, n, hits, ndistinct, lds_keys, bucket_mul);
}

// One WAVE per genome, four genomes per workgroup: the first sort launch of a batch whose genomes are EXPECTED to sample at
// most a few dozen k-mers (plasmids, viral genomes, contigs of a few kbp: 400 000 genomes of 2 kbp have 1.3 hashes each --
// there one 512-thread workgroup per genome, seven of whose eight waves leave at once, cost 0.74 ms against 2.25 ms for
// the k-mer kernel).  A genome with more than 64 raw hits is left to hg_launch_sort_unique_rest (skip_keys = 64).
__global__ __launch_bounds__(256) void sort_unique_wave_kernel(
    const hg_genome_meta *__restrict__ meta, uint64_t *__restrict__ hits, const uint32_t *__restrict__ cnt,
    uint32_t *__restrict__ ndistinct, uint32_t n_genomes, uint32_t *__restrict__ flags) {
  const uint32_t g = __builtin_amdgcn_readfirstlane(blockIdx.x * 4 + (threadIdx.x >> 6)), lane = threadIdx.x & 63;
  if (g >= n_genomes) return;
  uint32_t n = cnt[g];
  const uint32_t cap = meta[g].hit_cap;
  if (flags) {
    const bool over = n > cap;
    if (over || n > SORT_LDS_MAX_KEYS) {
      if (lane == 0) atomicOr(flags, over ? HG_STEP_OVERFLOW : HG_STEP_LARGE_SET), ndistinct[g] = HG_NHASH_PENDING;
      return;
    }
  }
  if (n > cap) n = cap;  // overflow is reported by the host from cnt[]
  if (n > 64) return;
  if (n <= 1) {
    if (lane == 0) ndistinct[g] = n;
    return;
  }
  sort_unique_wave(hits + meta[g].hit_off, n, lane, ndistinct + g);
}

// grid: SORT_WG genomes per workgroup.  The genomes whose raw count is in (skip_keys, lds_keys] -- what a launch of
// sort_unique_kernel with skip_keys of LDS left out -- are picked out of the counters by the workgroup itself and sorted
// one after the other (rare by construction: skip_keys is 1.125 times the expected count, or last run's largest).
__global__ __launch_bounds__(SORT_WG) void sort_unique_rest_kernel(
    const hg_genome_meta *__restrict__ meta, uint64_t *__restrict__ hits, const uint32_t *__restrict__ cnt,
    uint32_t *__restrict__ ndistinct, uint32_t n_genomes, uint32_t skip_keys, uint32_t lds_keys, uint64_t bucket_mul) {
  __shared__ uint32_t s_list[SORT_WG], s_n;
  if (threadIdx.x == 0) s_n = 0;
  __syncthreads();
  const uint32_t g = blockIdx.x * SORT_WG + threadIdx.x;
  if (g < n_genomes) {
    const uint32_t c = cnt[g];
    if (c > skip_keys && c <= lds_keys && c <= meta[g].hit_cap) s_list[atomicAdd(&s_n, 1u)] = g;
  }
  __syncthreads();
  const uint32_t todo = s_n;
  for (uint32_t i = 0; i < todo; ++i) {
    const uint32_t gi = s_list[i];
    const hg_genome_meta gm = meta[gi];
    sort_unique_one<true>(gi, gm, cnt[gi], hits, ndistinct, lds_keys, bucket_mul);
    __syncthreads();
  }
}

__global__ __launch_bounds__(256) void sketch_finish_kernel(const uint32_t *__restrict__ ndistinct, uint32_t *__restrict__ nhash,
                                                            uint32_t n_genomes, const uint32_t *__restrict__ flags,
                                                            volatile uint32_t *h_slot, uint32_t seq) {
  const uint32_t g = blockIdx.x * 256 + threadIdx.x;
  if (g < n_genomes) nhash[g] = ndistinct[g];
  if (blockIdx.x == 0 && threadIdx.x == 0) {
    h_slot[0] = *flags;
    __threadfence_system();
    h_slot[1] = seq;
    __threadfence_system();
  }
}

// ---- large hash sets: bucket by value, sort + unique each bucket in LDS -----------------------------------
constexpr uint32_t BK_WG = 256;
__device__ __forceinline__ uint32_t bucket_of(uint64_t h, const hg_bucket_job &job) {
  const uint32_t b = (uint32_t)__umul64hi(h, job.mul);  // monotone in h
  return b < job.P ? b : job.P - 1;
}

// grid: key chunks.  bcount[bucket] += the chunk's keys of that bucket.  A chunk of 4 096 keys of a genome with up to
// BK_PRIV_MAX buckets counts in LDS first and adds its non-zero counters once (a global atomic per KEY on the genome's few
// counters ran at 0.38 TB/s of keys: 1.0 ms for 50 M keys); beyond that a key hits a bucket less than four times per chunk
// and goes straight to the global counter (no value returned: the waves do not wait).
constexpr uint32_t BK_PRIV_MAX = 2048, BK_KPT = HG_BUCKET_CHUNK / BK_WG;
static_assert(HG_BUCKET_CHUNK % BK_WG == 0, "whole keys per thread");
__global__ __launch_bounds__(BK_WG) void bucket_count_kernel(const hg_bucket_job *__restrict__ jobs,
                                                             const uint32_t *__restrict__ chunk_job,
                                                             const uint64_t *__restrict__ hits,
                                                             uint32_t *__restrict__ bcount) {
  __shared__ uint32_t s_h[BK_PRIV_MAX];
  const hg_bucket_job job = jobs[chunk_job[blockIdx.x]];
  const uint32_t k0 = (blockIdx.x - job.chunk_first) * HG_BUCKET_CHUNK;
  const uint32_t k1 = k0 + HG_BUCKET_CHUNK < job.n ? k0 + HG_BUCKET_CHUNK : job.n;
  if (job.P > BK_PRIV_MAX) {  // workgroup-uniform
    for (uint32_t i = k0 + threadIdx.x; i < k1; i += BK_WG)
      atomicAdd(&bcount[job.bucket_first + bucket_of(hits[job.hit_off + i], job)], 1u);
    return;
  }
  for (uint32_t b = threadIdx.x; b < job.P; b += BK_WG) s_h[b] = 0;
  __syncthreads();
  for (uint32_t i = k0 + threadIdx.x; i < k1; i += BK_WG) atomicAdd(&s_h[bucket_of(hits[job.hit_off + i], job)], 1u);
  __syncthreads();
  for (uint32_t b = threadIdx.x; b < job.P; b += BK_WG) {
    const uint32_t v = s_h[b];
    if (v) atomicAdd(&bcount[job.bucket_first + b], v);
  }
}

// grid: jobs.  out[b] = exclusive prefix of in[b] over the job's buckets; optionally the total per genome
__global__ __launch_bounds__(SORT_WG) void bucket_scan_kernel(const hg_bucket_job *__restrict__ jobs,
                                                              const uint32_t *__restrict__ in,
                                                              uint32_t *__restrict__ out,
                                                              uint32_t *__restrict__ total_per_genome) {
  __shared__ uint32_t s_scan[SORT_WG / 64 + 1];
  const hg_bucket_job job = jobs[blockIdx.x];
  uint32_t run = 0;
  for (uint32_t b0 = 0; b0 < job.P; b0 += SORT_WG) {
    const uint32_t b = b0 + threadIdx.x;
    const uint32_t v = b < job.P ? in[job.bucket_first + b] : 0u;
    uint32_t total;
    const uint32_t pre = block_excl_scan(v, s_scan, &total);
    if (b < job.P) out[job.bucket_first + b] = run + pre;
    run += total;
  }
  if (total_per_genome && threadIdx.x == 0) total_per_genome[job.genome] = run;
}

// grid: key chunks.  Every key moves to its bucket's range of the scratch buffer.  With up to BK_PRIV_MAX buckets the chunk
// ranks its keys per bucket in LDS (returning LDS atomics), reserves ONE run per non-empty bucket (a returning global atomic
// per bucket instead of per key: 2.1 -> ... ms for 50 M keys) and writes the keys, held in registers meanwhile, into the runs.
__global__ __launch_bounds__(BK_WG) void bucket_scatter_kernel(const hg_bucket_job *__restrict__ jobs,
                                                               const uint32_t *__restrict__ chunk_job,
                                                               const uint64_t *__restrict__ hits,
                                                               const uint32_t *__restrict__ bstart,
                                                               uint32_t *__restrict__ bcursor,
                                                               uint64_t *__restrict__ tmp) {
  __shared__ uint32_t s_h[BK_PRIV_MAX];
  const hg_bucket_job job = jobs[chunk_job[blockIdx.x]];
  const uint32_t k0 = (blockIdx.x - job.chunk_first) * HG_BUCKET_CHUNK;
  const uint32_t k1 = k0 + HG_BUCKET_CHUNK < job.n ? k0 + HG_BUCKET_CHUNK : job.n;
  if (job.P > BK_PRIV_MAX) {  // workgroup-uniform
    for (uint32_t i = k0 + threadIdx.x; i < k1; i += BK_WG) {
      const uint64_t h = hits[job.hit_off + i];
      const uint32_t gb = job.bucket_first + bucket_of(h, job);
      const uint32_t pos = atomicAdd(&bcursor[gb], 1u);
      tmp[job.hit_off + bstart[gb] + pos] = h;
    }
    return;
  }
  for (uint32_t b = threadIdx.x; b < job.P; b += BK_WG) s_h[b] = 0;
  __syncthreads();
  uint64_t kk[BK_KPT];
  uint32_t bb[BK_KPT], rr[BK_KPT];
#pragma unroll
  for (uint32_t u = 0; u < BK_KPT; ++u) {
    const uint32_t i = k0 + threadIdx.x + u * BK_WG;
    if (i < k1) {
      kk[u] = hits[job.hit_off + i];
      bb[u] = bucket_of(kk[u], job);
      rr[u] = atomicAdd(&s_h[bb[u]], 1u);  // rank among the chunk's keys of that bucket
    }
  }
  __syncthreads();
  for (uint32_t b = threadIdx.x; b < job.P; b += BK_WG) {  // count -> where the chunk's run of bucket b starts in the scratch copy
    const uint32_t v = s_h[b], gb = job.bucket_first + b;
    if (v) s_h[b] = bstart[gb] + atomicAdd(&bcursor[gb], v);
  }
  __syncthreads();
#pragma unroll
  for (uint32_t u = 0; u < BK_KPT; ++u)
    if (k0 + threadIdx.x + u * BK_WG < k1) tmp[job.hit_off + s_h[bb[u]] + rr[u]] = kk[u];
}

// grid: buckets.  Sort + unique in LDS; the distinct keys go back to the start of the bucket's scratch range.
// A bucket with more keys than LDS holds (only possible when duplicates pile up: the map is balanced for
// distinct hashes) is first de-duplicated through an LDS hash set; if even its distinct keys do not fit the
// job is flagged and the caller sorts that genome in place instead.
__global__ __launch_bounds__(SORT_WG) void bucket_sort_kernel(const hg_bucket_job *__restrict__ jobs,
                                                              const uint32_t *__restrict__ bucket_job,
                                                              const uint32_t *__restrict__ bcount,
                                                              const uint32_t *__restrict__ bstart,
                                                              uint64_t *__restrict__ tmp,
                                                              uint32_t *__restrict__ bdist,
                                                              uint32_t *__restrict__ fail, uint32_t cap_keys) {
  // cap_keys (a power of two, <= SORT_LDS_MAX_KEYS): keys the launch's LDS holds -- 12 bytes each, keys + counters.  The
  // launcher sizes it to four times the bucket size the plan aims at: 48 KiB for 512-1 024 expected keys, three workgroups
  // per CU (with the full 96 KiB in every launch one workgroup per CU sorted 1 500 keys at a time).
  const uint32_t HSET_SLOTS = cap_keys, HSET_MAX = HSET_SLOTS / 4 * 3;
  extern __shared__ __attribute__((aligned(16))) uint64_t s_keys[];
  __shared__ uint32_t s_scan[SORT_WG / 64 + 1];
  __shared__ uint32_t s_distinct;
  const uint32_t gb = blockIdx.x, j = bucket_job[gb], tid = threadIdx.x;
  const hg_bucket_job job = jobs[j];
  const uint32_t n = bcount[gb];
  uint64_t *base = tmp + job.hit_off + bstart[gb];
  if (n == 0) {
    if (tid == 0) bdist[gb] = 0;
    return;
  }
  if (n > cap_keys) {
    for (uint32_t i = tid; i < HSET_SLOTS; i += SORT_WG) s_keys[i] = ~0ull;  // no hash equals ~0 (h < threshold)
    if (tid == 0) s_distinct = 0;
    __syncthreads();
    for (uint32_t i = tid; i < n; i += SORT_WG) {
      const uint64_t h = base[i];
      uint32_t slot = (uint32_t)((h * 0x9E3779B97F4A7C15ull) >> 40) & (HSET_SLOTS - 1);
      for (;;) {
        if (s_distinct > HSET_MAX) break;  // hopeless: flagged below
        const uint64_t old = atomicCAS(reinterpret_cast<unsigned long long *>(&s_keys[slot]), ~0ull, (unsigned long long)h);
        if (old == ~0ull) {
          atomicAdd(&s_distinct, 1u);
          break;
        }
        if (old == h) break;
        slot = (slot + 1) & (HSET_SLOTS - 1);
      }
    }
    __syncthreads();
    if (s_distinct > HSET_MAX) {
      if (tid == 0) fail[j] = 1u, bdist[gb] = 0;
      return;
    }
    bitonic_sort(s_keys, HSET_SLOTS, tid, SORT_WG);  // empty slots (~0) sort to the end
    const uint32_t d = s_distinct;
    for (uint32_t i = tid; i < d; i += SORT_WG) base[i] = s_keys[i];
    if (tid == 0) bdist[gb] = d;
    return;
  }
  const uint32_t n2 = next_pow2(n);
  bool sorted = false;  // workgroup-uniform
  if (n2 >= (uint32_t)SORT_WG) {
    // The bucket's keys are uniform over its value range: the same counting sort as sort_unique_kernel's, one level down --
    // sub-bucket = the top bits of the FRACTION of h * mul (its integer part is the bucket; the fraction grows with h inside
    // it), 0.5-1 keys per sub-bucket, ranks by returning LDS atomics, one scan, one scatter, an insertion pass per thread.
    // Five passes over the keys instead of the bitonic network's 66 (2 048 keys): 3.4 -> ... ms for 50 M keys in 32 000 buckets.
    __shared__ uint32_t s_over;
    uint32_t *s_bk = reinterpret_cast<uint32_t *>(s_keys + cap_keys);  // n2 counters, then sub-bucket starts
    const uint32_t shift = 64u - (uint32_t)__builtin_ctz(n2), per = n2 / SORT_WG;
    for (uint32_t i = tid; i < n2; i += SORT_WG) s_bk[i] = 0;
    if (tid == 0) s_over = 0;
    __syncthreads();
    uint64_t kk[SORT_KPT];
    uint32_t bb[SORT_KPT], rr[SORT_KPT];
#pragma unroll
    for (uint32_t u = 0; u < SORT_KPT; ++u) {
      const uint32_t i = tid + u * SORT_WG;
      if (i < n) {
        kk[u] = base[i];
        // (keys the bucket map clamped into the last bucket -- integer part >= P -- have no usable fraction: last sub-bucket)
        const bool clamped = (uint32_t)__umul64hi(kk[u], job.mul) >= job.P;
        bb[u] = clamped ? n2 - 1 : (uint32_t)((kk[u] * job.mul) >> shift);
        rr[u] = atomicAdd(&s_bk[bb[u]], 1u);
      }
    }
    __syncthreads();
    {
      uint32_t c[SORT_KPT], sum = 0, mx = 0;
#pragma unroll
      for (uint32_t q = 0; q < SORT_KPT; ++q)
        if (q < per) c[q] = s_bk[tid * per + q], sum += c[q], mx = c[q] > mx ? c[q] : mx;
      if (mx > SORT_BUCKET_LIMIT) s_over = 1u;  // (same value from every writer)
      uint32_t total;
      uint32_t run0 = block_excl_scan(sum, s_scan, &total);
#pragma unroll
      for (uint32_t q = 0; q < SORT_KPT; ++q)
        if (q < per) s_bk[tid * per + q] = run0, run0 += c[q];
    }
    __syncthreads();
#pragma unroll
    for (uint32_t u = 0; u < SORT_KPT; ++u)
      if (tid + u * SORT_WG < n) s_keys[s_bk[bb[u]] + rr[u]] = kk[u];
    __syncthreads();
    sorted = s_over == 0u;
    if (sorted) {
      for (uint32_t q = 0; q < per; ++q) {  // order the keys inside each of this thread's sub-buckets
        const uint32_t b = tid * per + q, lo = s_bk[b], hi = b + 1 < n2 ? s_bk[b + 1] : n;
        for (uint32_t i = lo + 1; i < hi; ++i) {
          const uint64_t v = s_keys[i];
          uint32_t j2 = i;
          while (j2 > lo && s_keys[j2 - 1] > v) s_keys[j2] = s_keys[j2 - 1], --j2;
          s_keys[j2] = v;
        }
      }
    } else {
      for (uint32_t i = n + tid; i < n2; i += SORT_WG) s_keys[i] = ~0ull;  // (piled-up duplicates: the network after all)
    }
    __syncthreads();
  } else {
    for (uint32_t i = tid; i < n2; i += SORT_WG) s_keys[i] = (i < n) ? base[i] : ~0ull;
    __syncthreads();
  }
  if (!sorted) bitonic_sort(s_keys, n2, tid, SORT_WG);
  uint32_t run = 0;
  for (uint32_t c0 = 0; c0 < n; c0 += SORT_WG) {
    const uint32_t i = c0 + tid;
    uint64_t v = 0;
    uint32_t keep = 0;
    if (i < n) {
      v = s_keys[i];
      keep = (i == 0 || v != s_keys[i - 1]) ? 1u : 0u;
    }
    uint32_t total;
    const uint32_t pos = block_excl_scan(keep, s_scan, &total);
    if (keep) base[run + pos] = v;
    run += total;
  }
  if (tid == 0) bdist[gb] = run;
}

// grid: buckets.  Distinct keys of the bucket -> their final place in the genome's hit region.
__global__ __launch_bounds__(BK_WG) void bucket_copy_kernel(const hg_bucket_job *__restrict__ jobs,
                                                            const uint32_t *__restrict__ bucket_job,
                                                            const uint32_t *__restrict__ bstart,
                                                            const uint32_t *__restrict__ bdist,
                                                            const uint32_t *__restrict__ bout,
                                                            const uint32_t *__restrict__ fail,
                                                            const uint64_t *__restrict__ tmp,
                                                            uint64_t *__restrict__ hits) {
  const uint32_t gb = blockIdx.x;
  if (fail[bucket_job[gb]]) return;  // the genome's raw keys must survive for the in-place sort
  const hg_bucket_job job = jobs[bucket_job[gb]];
  const uint64_t *src = tmp + job.hit_off + bstart[gb];
  uint64_t *dst = hits + job.hit_off + bout[gb];
  for (uint32_t i = threadIdx.x; i < bdist[gb]; i += BK_WG) dst[i] = src[i];
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
// (majority and three-way xor as ONE v_bitop3_b32 per 32-bit half each -- truth tables 0xE8 and 0x96 with a = 0xF0,
// b = 0xCC, c = 0xAA --: 4 instructions per adder instead of the 10 the compiler makes of and / or / xor)
__device__ __forceinline__ void csa(uint64_t &hi, uint64_t &lo, uint64_t a, uint64_t b, uint64_t c) {
  const uint32_t a0 = (uint32_t)a, a1 = (uint32_t)(a >> 32), b0 = (uint32_t)b, b1 = (uint32_t)(b >> 32);
  const uint32_t c0 = (uint32_t)c, c1 = (uint32_t)(c >> 32);
  const uint32_t h0 = __builtin_amdgcn_bitop3_b32(a0, b0, c0, 0xE8), h1 = __builtin_amdgcn_bitop3_b32(a1, b1, c1, 0xE8);
  const uint32_t l0 = __builtin_amdgcn_bitop3_b32(a0, b0, c0, 0x96), l1 = __builtin_amdgcn_bitop3_b32(a1, b1, c1, 0x96);
  hi = (uint64_t)h0 | ((uint64_t)h1 << 32);
  lo = (uint64_t)l0 | ((uint64_t)l1 << 32);
}

constexpr int HI_PLANES = 10;  // counts up to 15 + 16*1023 per flush window

// One workgroup per genome.  hv_d/64 words are spread over lanes; when hv_d/64 < 64*ENC_WAVES
// several waves share a word and split the hashes.
// SPLIT = false: one workgroup per genome, writes the hypervector.  Genomes with more than `split_over`
// hashes are left to the split launch (split_over = ~0u: none are).
// SPLIT = true : one workgroup per (genome, slab of HG_ENC_SLAB hashes) item; the per-dimension counts are
// added into accum[slot][word * 64 + bit] and encode_finalize_kernel turns them into the hypervector --
// a 3 Gbp genome (2 M hashes) otherwise keeps one workgroup busy for 30 ms while hashing it takes 8.
template <bool SPLIT>
__global__ __launch_bounds__(ENC_WG) void encode_kernel(
    const hg_genome_meta *__restrict__ meta, const uint64_t *__restrict__ hits,
    const uint32_t *__restrict__ ndistinct, uint32_t hv_d, uint32_t layout,
    int16_t *__restrict__ hv_out, int32_t *__restrict__ norm2_out, uint32_t split_over,
    const uint2 *__restrict__ items, uint32_t *__restrict__ accum, uint32_t wave_max) {
  extern __shared__ __attribute__((aligned(16))) uint32_t s_cnt[];  // [64][n_words + 1]
  __shared__ int32_t s_red[ENC_WAVES];
  const uint32_t g = SPLIT ? items[blockIdx.x].x : blockIdx.x;
  const hg_genome_meta gm = meta[g];
  const uint32_t n_all = ndistinct[g];
  if (!SPLIT && (n_all > split_over || n_all <= wave_max || n_all == HG_NHASH_PENDING)) return;  // split launch / encode_wave_kernel / left to the redo
  const uint32_t slab = SPLIT ? (items[blockIdx.x].y & 0xffffu) : 0u, slot = SPLIT ? (items[blockIdx.x].y >> 16) : 0u;
  const uint32_t h0 = SPLIT ? slab * HG_ENC_SLAB : 0u;
  if (SPLIT && h0 >= n_all) return;  // the plan was made from the raw (pre-unique) count
  const uint32_t n = SPLIT ? (n_all - h0 < HG_ENC_SLAB ? n_all - h0 : HG_ENC_SLAB) : n_all;  // hashes of this workgroup
  const uint64_t *__restrict__ hs = hits + gm.hit_off + h0;
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

  if (SPLIT) {
    uint32_t *__restrict__ a = accum + (size_t)slot * hv_d;
    for (uint32_t i = tid; i < n_words * 64; i += ENC_WG) {
      const uint32_t w = i >> 6, j = i & 63, c = s_cnt[j * stride + w];
      if (c) atomicAdd(&a[i], c);  // no value returned: nothing waits for it
    }
    return;
  }
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

typedef short short2v __attribute__((ext_vector_type(2)));

// The lane's 64 bit-column counts, held bit-sliced in planes pl[0..P) (count < 2^P), as its 64 outputs 2 * count - n in the
// AVX2 dimension order, two per dword, + their squares into acc.  Output position p of the 64-block holds bit
// j = 16 * (p % 4) + p / 4 (src/hd.rs:14-92), so the pair (2 m, 2 m + 1) holds bits (b, b + 16) of ONE 32-bit half of the
// planes, b = m / 2 of the low half for even m, of the high half for odd m: one shift and one v_and_or per plane moves
// both bits of plane k to bit k of their 16-bit fields -- 2 P instructions per pair, then packed 16-bit arithmetic
// (v_pk_lshlrev_b16, v_pk_sub_i16, v_dot2c_i32_i16).  (The generic form below extracts every bit of every plane on its own:
// 2 900 instructions per lane for P = 14 against 32 (2 P + 3).)
template <int P>
__device__ __forceinline__ void expand_pairs_avx2(const uint64_t *pl, uint32_t n, uint32_t (&packed)[32], uint32_t &acc) {
  const short2v nv = {(short)(uint16_t)n, (short)(uint16_t)n};
#pragma unroll
  for (int m = 0; m < 32; ++m) {
    uint32_t c = 0;
#pragma unroll
    for (int k = 0; k < P; ++k) {
      const int t = m >> 1;
      const uint32_t xw = (m & 1) ? (uint32_t)(pl[k] >> 32) : (uint32_t)pl[k];
      const uint32_t sh = t >= k ? xw >> (t - k) : xw << (k - t);
      c |= sh & (0x00010001u << k);
    }
    const short2v cv = __builtin_bit_cast(short2v, c), v = cv + cv - nv;  // i16 wrapping, like the reference's i16 sums
    acc = (uint32_t)__builtin_amdgcn_sdot2(v, v, (int)acc, false);  // (no clamp: wraps like the i32 sum)
    packed[m] = __builtin_bit_cast(uint32_t, v);
  }
}

// One WAVE per genome (four genomes per workgroup) for hash sets of at most HG_ENC_WAVE_MAX hashes -- every
// ordinary genome.  Lane i owns word i of the random stream (hv_d / 64 <= 64 words per pass), so the 64 bit
// columns of that word are counted entirely inside the lane: bit-sliced carry-save planes over all hashes,
// expanded once at the end straight into the output row.  No atomics, no barrier; the
// eight-waves-per-genome kernel above (LDS counters, one flush per wave and word) remains for larger sets.
// (Which genomes it takes: hg_launch_encode.)  A genome of a few kbp has 1-30 hashes: there the expansion IS the kernel,
// and it is specialised on the number of planes the counts can occupy (4 for n < 16).
__global__ __launch_bounds__(256) void encode_wave_kernel(
    const hg_genome_meta *__restrict__ meta, const uint64_t *__restrict__ hits,
    const uint32_t *__restrict__ ndistinct, uint32_t n_genomes, uint32_t hv_d, uint32_t layout,
    int16_t *__restrict__ hv_out, int32_t *__restrict__ norm2_out, uint32_t wave_max) {
  const uint32_t lane = threadIdx.x & 63;
  const uint32_t g = __builtin_amdgcn_readfirstlane(blockIdx.x * 4 + (threadIdx.x >> 6));
  if (g >= n_genomes) return;
  const uint32_t n = ndistinct[g];
  if (n > wave_max) return;  // encode_kernel handles this genome
  const uint64_t *__restrict__ hs = hits + meta[g].hit_off;
  const uint32_t n_words = hv_d / 64;
  int16_t *__restrict__ out = hv_out + (size_t)g * hv_d;
  const bool vec_ok = (hv_d % 8 == 0) && ((reinterpret_cast<uintptr_t>(hv_out) & 15) == 0);
  // A lane holds 128 consecutive bytes of the row: stored from the registers, every store instruction of the wave touches 64
  // different 128-byte lines, 16 bytes of each.  For the shortest sets (n < 16: a genome of a few kbp), where the 8 KB row
  // IS the genome's cost, the pass goes through the wave's 8 KB of LDS instead -- lane l's q-th 16 bytes at slot
  // 8 l + (q ^ (l & 7)) -- and leaves with the lanes on consecutive addresses: 400 000 sets of 1-2 hashes 1.16 -> 0.78 ms
  // (4.2 TB/s of rows).  Taken for n < 64 as well, the kernel needs 181 registers instead of 121 and every larger set pays
  // (0.34 -> 0.46 ms at 33 hashes, 1.46 -> 2.14 at 3 300): not taken.
  __shared__ uint4 s_rows[4][512];
  uint4 *const s_row = s_rows[threadIdx.x >> 6];
  const bool via_lds = vec_ok && n < 16 && layout == HG_LAYOUT_AVX2;  // wave-uniform
  uint32_t acc = 0;
  for (uint32_t grp = 0; grp * 64 < n_words; ++grp) {
    const uint32_t w = grp * 64 + lane;
    const uint64_t off = (uint64_t)(w + 1) * WY_INC;
    uint64_t pl[4 + HI_PLANES];  // ones, twos, fours, eights, then the 16s planes
#pragma unroll
    for (int p = 0; p < 4 + HI_PLANES; ++p) pl[p] = 0;
    for (uint32_t b0 = 0; b0 < n; b0 += 16) {
      // (the sixteen hashes are wave-uniform scalar loads, all issued before the first is used: the index is clamped
      // instead of the load being skipped, which would put every load of a short set behind its own branch and wait)
      uint64_t h[16], x[16];
#pragma unroll
      for (int t = 0; t < 16; ++t) h[t] = hs[b0 + t < n ? b0 + t : n - 1];
#pragma unroll
      for (int t = 0; t < 16; ++t) x[t] = (b0 + t < n) ? wy_word(h[t], off) : 0ull;
      uint64_t twosA, twosB, foursA, foursB, eightsA, eightsB, sixteens;
      csa(twosA, pl[0], pl[0], x[0], x[1]);
      csa(twosB, pl[0], pl[0], x[2], x[3]);
      csa(foursA, pl[1], pl[1], twosA, twosB);
      csa(twosA, pl[0], pl[0], x[4], x[5]);
      csa(twosB, pl[0], pl[0], x[6], x[7]);
      csa(foursB, pl[1], pl[1], twosA, twosB);
      csa(eightsA, pl[2], pl[2], foursA, foursB);
      csa(twosA, pl[0], pl[0], x[8], x[9]);
      csa(twosB, pl[0], pl[0], x[10], x[11]);
      csa(foursA, pl[1], pl[1], twosA, twosB);
      csa(twosA, pl[0], pl[0], x[12], x[13]);
      csa(twosB, pl[0], pl[0], x[14], x[15]);
      csa(foursB, pl[1], pl[1], twosA, twosB);
      csa(eightsB, pl[2], pl[2], foursA, foursB);
      csa(sixteens, pl[3], pl[3], eightsA, eightsB);
      if (n >= 16) {  // (wave-uniform; fewer than 16 hashes never carry out of the eights)
        uint64_t carry = sixteens;
#pragma unroll
        for (int p = 0; p < HI_PLANES; ++p) {
          const uint64_t t = pl[4 + p] & carry;
          pl[4 + p] ^= carry;
          carry = t;
        }
      }
    }
    if (via_lds) {  // wave-uniform
      if (w < n_words) {
        uint32_t packed[32];
        expand_pairs_avx2<4>(pl, n, packed, acc);
#pragma unroll
        for (int q = 0; q < 8; ++q)
          s_row[8 * lane + ((uint32_t)q ^ (lane & 7u))] = make_uint4(packed[4 * q], packed[4 * q + 1], packed[4 * q + 2], packed[4 * q + 3]);
      }
      // (the wave's own LDS operations complete in order; the fences keep the compiler from moving them across each other)
      __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
      __builtin_amdgcn_wave_barrier();
      __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
      const uint32_t pass_words = n_words - grp * 64 < 64u ? n_words - grp * 64 : 64u;
      uint4 *dst = reinterpret_cast<uint4 *>(out + (size_t)grp * 64 * 64);
#pragma unroll 2
      for (uint32_t sidx = 0; sidx < 8; ++sidx) {
        const uint32_t i = sidx * 64 + lane, l = i >> 3, q = i & 7u;
        if (l < pass_words) {  // (written once, read by nobody on this device soon: past the caches)
          typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
          const uint4 v = s_row[8 * l + (q ^ (l & 7u))];
          __builtin_nontemporal_store(u32x4{v.x, v.y, v.z, v.w}, reinterpret_cast<u32x4 *>(dst + i));
        }
      }
      __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");  // (the next pass writes the same slots)
      __builtin_amdgcn_wave_barrier();
    } else if (w < n_words) {
      uint32_t packed[32];
      if (layout == HG_LAYOUT_AVX2) {  // wave-uniform, and so is the choice of the plane count
        if (n < 16) expand_pairs_avx2<4>(pl, n, packed, acc);
        else if (n < 64) expand_pairs_avx2<6>(pl, n, packed, acc);
        else if (n < 256) expand_pairs_avx2<8>(pl, n, packed, acc);
        else expand_pairs_avx2<4 + HI_PLANES>(pl, n, packed, acc);
      } else {  // scalar order: output position p holds bit p
#pragma unroll
        for (int p = 0; p < 64; ++p) {
          uint32_t c = 0;
#pragma unroll
          for (int q = 0; q < 4 + HI_PLANES; ++q) c |= (uint32_t)((pl[q] >> p) & 1) << q;
          const int16_t v = (int16_t)(uint16_t)(2u * c - n);
          acc += (uint32_t)((int32_t)v * (int32_t)v);
          if (p & 1) packed[p >> 1] |= (uint32_t)(uint16_t)v << 16;
          else packed[p >> 1] = (uint32_t)(uint16_t)v;
        }
      }
      int16_t *dst = out + (size_t)w * 64;
      if (vec_ok) {
#pragma unroll
        for (int q = 0; q < 8; ++q)  // (NOT past the caches like the LDS pass above: a lane's eight 16-byte stores lie in ONE 128-byte line the
                                     // L2 puts together -- written non-temporally they reach HBM as partial lines, 0.33 -> 1.68 ms at 33 hashes)
          reinterpret_cast<uint4 *>(dst)[q] = make_uint4(packed[4 * q], packed[4 * q + 1], packed[4 * q + 2], packed[4 * q + 3]);
      } else {
#pragma unroll
        for (int q = 0; q < 32; ++q) dst[2 * q] = (int16_t)(packed[q] & 0xffffu), dst[2 * q + 1] = (int16_t)(packed[q] >> 16);
      }
    }
  }
  // dimensions past the last whole 64-block carry no random bit: count 0
  for (uint32_t d = n_words * 64 + lane; d < hv_d; d += 64) {
    const int16_t v = (int16_t)(uint16_t)(0u - n);
    out[d] = v;
    acc += (uint32_t)((int32_t)v * (int32_t)v);
  }
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) acc += __shfl_down(acc, o);
  if (lane == 0) norm2_out[g] = (int32_t)acc;
}

// grid: split genomes.  accum[slot][word * 64 + bit] -> hv (layout, i16 wrapping) + norm
__global__ __launch_bounds__(ENC_WG) void encode_finalize_kernel(const uint32_t *__restrict__ genomes,
                                                                 const uint32_t *__restrict__ ndistinct,
                                                                 const uint32_t *__restrict__ accum, uint32_t hv_d,
                                                                 uint32_t layout, int16_t *__restrict__ hv_out,
                                                                 int32_t *__restrict__ norm2_out) {
  __shared__ int32_t s_red[ENC_WAVES];
  const uint32_t g = genomes[blockIdx.x], n = ndistinct[g], tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const uint32_t *__restrict__ a = accum + (size_t)blockIdx.x * hv_d;
  int16_t *__restrict__ out = hv_out + (size_t)g * hv_d;
  const uint32_t d_full = (hv_d / 64) * 64;
  uint32_t acc = 0;
  for (uint32_t d = tid; d < hv_d; d += ENC_WG) {
    uint32_t c = 0;
    if (d < d_full) {
      const uint32_t w = d >> 6, pos = d & 63;
      const uint32_t j = (layout == HG_LAYOUT_AVX2) ? (16 * (pos & 3) + (pos >> 2)) : pos;
      c = a[w * 64 + j];
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
    uint32_t sum = 0;
    for (int w = 0; w < ENC_WAVES; ++w) sum += (uint32_t)s_red[w];
    norm2_out[g] = (int32_t)sum;
  }
}

}  // namespace

// hipFuncSetAttribute is per device: remember per device (one context per GPU may live in one process)
static bool attr_done_on_this_device(std::atomic<uint64_t> &mask, bool set) {
  int dev = 0;
  if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev > 63) return false;
  if (set) mask.fetch_or(1ull << dev);
  return (mask.load() >> dev) & 1;
}

static hipError_t sort_lds_attr() {
  static std::atomic<uint64_t> done{0};
  if (attr_done_on_this_device(done, false)) return hipSuccess;
  // (keys + the counting sort's counters)
  hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void *>(&sort_unique_kernel<true>),
                                     hipFuncAttributeMaxDynamicSharedMemorySize, SORT_LDS_BYTES_MAX);
  if (e == hipSuccess)
    e = hipFuncSetAttribute(reinterpret_cast<const void *>(&bucket_sort_kernel),
                            hipFuncAttributeMaxDynamicSharedMemorySize, SORT_LDS_BYTES_MAX);
  if (e == hipSuccess)
    e = hipFuncSetAttribute(reinterpret_cast<const void *>(&sort_unique_rest_kernel),
                            hipFuncAttributeMaxDynamicSharedMemorySize, SORT_LDS_BYTES_MAX);
  if (e == hipSuccess) attr_done_on_this_device(done, true);
  return e;
}

uint32_t hg_sort_lds_keys(uint32_t max_cap) {
  uint32_t keys = 1;
  while (keys < max_cap && keys < SORT_LDS_MAX_KEYS) keys <<= 1;
  return keys;
}

// dynamic LDS of the LDS sort: the keys, and the bucket counters of the counting-sort fast path where it applies
static size_t sort_lds_bytes(uint32_t keys, uint64_t bucket_mul) {
  return (size_t)keys * sizeof(uint64_t) + ((bucket_mul && keys <= SORT_BUCKET_MAX_KEYS) ? (size_t)keys * sizeof(uint32_t) : 0);
}
// bucket_mul for hashes below `threshold` and `keys` buckets: ceil(keys * 2^64 / threshold) (0: no fast path)
static uint64_t sort_bucket_mul(uint32_t keys, uint64_t threshold) {
  if (threshold == 0 || keys > SORT_BUCKET_MAX_KEYS) return 0;
  const unsigned __int128 q = (((unsigned __int128)keys << 64) + threshold - 1) / threshold;
  return q > (unsigned __int128)UINT64_MAX ? UINT64_MAX : (uint64_t)q;
}

hipError_t hg_launch_sort_unique_todo(hipStream_t st, const hg_genome_meta *d_meta, const uint32_t *d_todo, uint32_t n_todo,
                                      uint64_t *d_hits, const uint32_t *d_cnt, uint32_t *d_ndistinct, uint32_t max_cap,
                                      uint64_t threshold) {
  if (n_todo == 0) return hipSuccess;
  const uint32_t keys = hg_sort_lds_keys(max_cap);
  const uint64_t bucket_mul = sort_bucket_mul(keys, threshold);
  hipError_t e = sort_lds_attr();
  if (e != hipSuccess) return e;
  hipLaunchKernelGGL((sort_unique_kernel<true>), dim3(n_todo), dim3(SORT_WG), sort_lds_bytes(keys, bucket_mul), st, d_meta,
                     d_hits, d_cnt, d_ndistinct, keys, d_todo, bucket_mul, (uint32_t *)nullptr);
  return hipGetLastError();
}

hipError_t hg_launch_sort_unique(hipStream_t st, const hg_genome_meta *d_meta, uint32_t n_genomes,
                                 uint64_t *d_hits, const uint32_t *d_cnt, uint32_t *d_ndistinct,
                                 uint32_t max_cap, uint64_t threshold, uint32_t *d_flags) {
  if (n_genomes == 0) return hipSuccess;
  const uint32_t keys = hg_sort_lds_keys(max_cap);
  const uint64_t bucket_mul = sort_bucket_mul(keys, threshold);
  const size_t lds = sort_lds_bytes(keys, bucket_mul);
  hipError_t e = sort_lds_attr();
  if (e != hipSuccess) return e;
  if (keys <= 64) {  // tiny sets: a wave per genome
    hipLaunchKernelGGL(sort_unique_wave_kernel, dim3((n_genomes + 3) / 4), dim3(256), 0, st, d_meta, d_hits, d_cnt, d_ndistinct,
                       n_genomes, d_flags);
    return hipGetLastError();
  }
  // genomes whose hit count exceeds the LDS budget are skipped here: the caller learns the counts and
  // runs hg_launch_sort_large / hg_launch_sort_inplace for them (or, with d_flags, reads the step's flag word)
  hipLaunchKernelGGL((sort_unique_kernel<true>), dim3(n_genomes), dim3(SORT_WG), lds, st, d_meta,
                     d_hits, d_cnt, d_ndistinct, keys, (const uint32_t *)nullptr, bucket_mul, d_flags);
  return hipGetLastError();
}

hipError_t hg_launch_sort_unique_rest(hipStream_t st, const hg_genome_meta *d_meta, uint32_t n_genomes, uint64_t *d_hits,
                                      const uint32_t *d_cnt, uint32_t *d_ndistinct, uint32_t done_cap, uint32_t max_cap,
                                      uint64_t threshold) {
  const uint32_t skip = hg_sort_lds_keys(done_cap), keys = hg_sort_lds_keys(max_cap);
  if (n_genomes == 0 || skip >= keys) return hipSuccess;
  const uint64_t bucket_mul = sort_bucket_mul(keys, threshold);
  hipError_t e = sort_lds_attr();
  if (e != hipSuccess) return e;
  hipLaunchKernelGGL(sort_unique_rest_kernel, dim3((n_genomes + SORT_WG - 1) / SORT_WG), dim3(SORT_WG),
                     sort_lds_bytes(keys, bucket_mul), st, d_meta, d_hits, d_cnt, d_ndistinct, n_genomes, skip, keys, bucket_mul);
  return hipGetLastError();
}

hipError_t hg_launch_sketch_finish(hipStream_t st, const uint32_t *d_ndistinct, uint32_t *d_nhash, uint32_t n_genomes,
                                   const uint32_t *d_flags, uint32_t *h_slot, uint32_t seq) {
  hipLaunchKernelGGL(sketch_finish_kernel, dim3((n_genomes + 255) / 256), dim3(256), 0, st, d_ndistinct, d_nhash, n_genomes,
                     d_flags, h_slot, seq);
  return hipGetLastError();
}

hipError_t hg_launch_sort_inplace(hipStream_t st, const hg_genome_meta *d_meta, const uint32_t *d_todo,
                                  uint32_t n_todo, uint64_t *d_hits, const uint32_t *d_cnt, uint32_t *d_ndistinct) {
  if (n_todo == 0) return hipSuccess;
  hipLaunchKernelGGL((sort_unique_kernel<false>), dim3(n_todo), dim3(SORT_WG), 0, st, d_meta, d_hits, d_cnt,
                     d_ndistinct, SORT_LDS_MAX_KEYS, d_todo, (uint64_t)0, (uint32_t *)nullptr);
  return hipGetLastError();
}

hipError_t hg_launch_sort_large(hipStream_t st, const hg_bucket_job *d_jobs, uint32_t n_jobs,
                                const uint32_t *d_chunk_job, uint32_t n_chunks, const uint32_t *d_bucket_job,
                                uint32_t n_buckets, uint32_t *d_bk, uint64_t *d_hits, uint64_t *d_tmp,
                                uint32_t *d_ndistinct, uint32_t bucket_cap_keys) {
  if (n_jobs == 0) return hipSuccess;
  uint32_t cap_keys = (uint32_t)SORT_WG;  // (a power of two: the counting sort deals n2 / SORT_WG sub-buckets to a thread)
  while (cap_keys < bucket_cap_keys && cap_keys < SORT_LDS_MAX_KEYS) cap_keys <<= 1;
  hipError_t e = sort_lds_attr();
  if (e != hipSuccess) return e;
  uint32_t *bcount = d_bk, *bstart = d_bk + n_buckets, *bcursor = d_bk + 2 * (size_t)n_buckets;
  uint32_t *bdist = d_bk + 3 * (size_t)n_buckets, *bout = d_bk + 4 * (size_t)n_buckets, *fail = d_bk + 5 * (size_t)n_buckets;
  if ((e = hipMemsetAsync(d_bk, 0, (5 * (size_t)n_buckets + n_jobs) * sizeof(uint32_t), st)) != hipSuccess) return e;
  hipLaunchKernelGGL(bucket_count_kernel, dim3(n_chunks), dim3(BK_WG), 0, st, d_jobs, d_chunk_job, d_hits, bcount);
  hipLaunchKernelGGL(bucket_scan_kernel, dim3(n_jobs), dim3(SORT_WG), 0, st, d_jobs, bcount, bstart, (uint32_t *)nullptr);
  hipLaunchKernelGGL(bucket_scatter_kernel, dim3(n_chunks), dim3(BK_WG), 0, st, d_jobs, d_chunk_job, d_hits, bstart,
                     bcursor, d_tmp);
  hipLaunchKernelGGL(bucket_sort_kernel, dim3(n_buckets), dim3(SORT_WG), (size_t)cap_keys * (sizeof(uint64_t) + sizeof(uint32_t)), st,
                     d_jobs, d_bucket_job, bcount, bstart, d_tmp, bdist, fail, cap_keys);
  hipLaunchKernelGGL(bucket_scan_kernel, dim3(n_jobs), dim3(SORT_WG), 0, st, d_jobs, bdist, bout, d_ndistinct);
  hipLaunchKernelGGL(bucket_copy_kernel, dim3(n_buckets), dim3(BK_WG), 0, st, d_jobs, d_bucket_job, bstart, bdist, bout,
                     fail, d_tmp, d_hits);
  return hipGetLastError();
}

static hipError_t encode_attr() {
  static std::atomic<uint64_t> done{0};
  if (attr_done_on_this_device(done, false)) return hipSuccess;
  hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void *>(&encode_kernel<false>),
                                     hipFuncAttributeMaxDynamicSharedMemorySize, 150 * 1024);
  if (e == hipSuccess)
    e = hipFuncSetAttribute(reinterpret_cast<const void *>(&encode_kernel<true>),
                            hipFuncAttributeMaxDynamicSharedMemorySize, 150 * 1024);
  if (e == hipSuccess) attr_done_on_this_device(done, true);
  return e;
}

hipError_t hg_launch_encode(hipStream_t st, const hg_genome_meta *d_meta, uint32_t n_genomes,
                            const uint64_t *d_hits, const uint32_t *d_ndistinct, uint32_t hv_d,
                            uint32_t layout, int16_t *d_hv, int32_t *d_norm2, const hg_encode_split *split,
                            uint32_t max_hashes) {
  if (n_genomes == 0) return hipSuccess;
  const size_t lds = (size_t)64 * (hv_d / 64 + 1) * sizeof(uint32_t);
  if (lds > 150 * 1024) return hipErrorInvalidValue;  // hv_d up to ~38k
  hipError_t e = encode_attr();
  if (e != hipSuccess) return e;
  const bool sp = split && split->n_items;
  // One wave per genome does less work per genome (no LDS counters, no per-wave flush) but runs a genome's
  // hashes serially: it takes everything when the batch alone fills the SIMDs several times over, and only the
  // tiny sets (where the eight-wave kernel is all overhead) otherwise.  1 000 x 3 333 hashes: 0.18 ms with eight
  // waves per genome, 0.39 ms with one; 100 000 x 20 hashes: 4.4 ms vs 0.7 ms.
  const uint32_t wave_max = n_genomes >= 8192 ? (uint32_t)HG_ENC_WAVE_MAX : 256u;
  hipLaunchKernelGGL(encode_wave_kernel, dim3((n_genomes + 3) / 4), dim3(256), 0, st, d_meta, d_hits, d_ndistinct, n_genomes,
                     hv_d, layout, d_hv, d_norm2, wave_max);
  if ((e = hipGetLastError()) != hipSuccess) return e;
  if (max_hashes > wave_max) {  // some genome may exceed what the wave kernel takes
    hipLaunchKernelGGL(encode_kernel<false>, dim3(n_genomes), dim3(ENC_WG), lds, st, d_meta, d_hits, d_ndistinct, hv_d,
                       layout, d_hv, d_norm2, sp ? (uint32_t)HG_ENC_SLAB : ~0u, (const uint2 *)nullptr, (uint32_t *)nullptr,
                       wave_max);
    if ((e = hipGetLastError()) != hipSuccess) return e;
  }
  if (!sp) return e;
  if ((e = hipMemsetAsync(split->d_accum, 0, (size_t)split->n_genomes * hv_d * sizeof(uint32_t), st)) != hipSuccess) return e;
  hipLaunchKernelGGL(encode_kernel<true>, dim3(split->n_items), dim3(ENC_WG), lds, st, d_meta, d_hits, d_ndistinct, hv_d,
                     layout, d_hv, d_norm2, ~0u, reinterpret_cast<const uint2 *>(split->d_items), split->d_accum, 0u);
  hipLaunchKernelGGL(encode_finalize_kernel, dim3(split->n_genomes), dim3(ENC_WG), 0, st, split->d_genomes, d_ndistinct,
                     split->d_accum, hv_d, layout, d_hv, d_norm2);
  return hipGetLastError();
}
