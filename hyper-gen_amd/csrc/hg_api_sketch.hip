// hg_api_sketch.hip -- the C ABI of include/hypergen.h, part 2: the sketch path -- batch plans, hash + sample -> sort / unique ->
// encode over the genomes of a batch (device-resident, host-fed with uploads and host-side 2-bit packing under them, one
// genome per call), page-locked read buffers.  What src/sketch.rs:35-56 and src/sketch_cuda.rs:79-166 do per file.
#include <algorithm>
#include <atomic>
#include <cctype>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <condition_variable>
#include <mutex>
#include <thread>
#include <vector>

#include <chrono>
#include <functional>
#include <memory>
#include <sched.h>

#include "hg_host.h"

#include "hg_internal.h"

// ---------------------------------------------------------------------------------------------
// sketch core: hash+sample -> sort/unique -> (optional) encode, all genomes of a batch
// ---------------------------------------------------------------------------------------------
namespace {

struct BatchPlan {
  uint32_t max_hits = ~0u;  // largest stored raw hit count of the batch (upper bound of the distinct counts)
  std::vector<std::pair<uint32_t, uint32_t>> big;  // (genome, stored raw hits) with more than HG_ENC_SLAB hits
  std::vector<hg_genome_meta> meta;
  std::vector<uint32_t> item_genome;
  uint64_t total_slots = 0;
  uint32_t max_cap = 0;
};

uint32_t round_cap(uint64_t cap) {
  if (cap > HG_SORT_LDS_MAX_KEYS) {  // in-place global sort needs a power of two
    uint64_t p = 1;
    while (p < cap) p <<= 1;
    cap = p;
  }
  return cap > 0xFFFFFFF0ull ? 0xFFFFFFF0u : (uint32_t)cap;
}

// want_caps: optional per-genome minimum capacities (retry after overflow)
hg_status make_plan(hg_ctx *c, const uint64_t *offsets, const uint64_t *lens, size_t n, uint32_t ksize,
                    uint64_t scaled, const std::vector<uint32_t> *want_caps, BatchPlan &pl, const uint64_t *mask_offs = nullptr) {
  const uint64_t item_starts = hg_kmer_item_starts(ksize);
  pl.meta.resize(n);
  pl.item_genome.clear();
  uint64_t slot = 0;
  uint32_t max_cap = 0;
  for (size_t g = 0; g < n; ++g) {
    if (offsets[g] & 3) return hg_fail(c, HG_ERR_INVALID, "genome offsets must be multiples of 4");
    hg_genome_meta &m = pl.meta[g];
    m.seq_off = offsets[g];
    m.n_bps = lens[g];
    m.mask_off = mask_offs ? mask_offs[g] : offsets[g] + (((lens[g] + 3) / 4 + 15) & ~(uint64_t)15);  // (read by the packed kernels only)
    const uint64_t n_starts = lens[g] >= ksize ? lens[g] - ksize + 1 : 0;
    uint64_t cap = n_starts / scaled * 2 + 1024;  // expected n_starts/scaled; sd ~ sqrt of that
    if (cap > n_starts) cap = n_starts;             // can never exceed the number of k-mers
    if (want_caps && (*want_caps)[g] > cap) cap = (*want_caps)[g];
    if (cap == 0) cap = 1;
    m.hit_cap = round_cap(cap);
    m.hit_off = slot;
    slot += m.hit_cap;
    max_cap = std::max(max_cap, m.hit_cap);
    const uint64_t n_items = (n_starts + item_starts - 1) / item_starts;
    if (pl.item_genome.size() + n_items > 0x7FFFFFFFull)
      return hg_fail(c, HG_ERR_UNSUPPORTED, "batch too large for one launch; split it");
    m.item_first = (uint32_t)pl.item_genome.size();
    pl.item_genome.insert(pl.item_genome.end(), (size_t)n_items, (uint32_t)g);
  }
  pl.total_slots = slot;
  pl.max_cap = max_cap;
  return HG_OK;
}

// Sorts + de-duplicates the genomes whose sampled hash count exceeds what one workgroup sorts in LDS.
// h_cnt: raw per-genome counters (host copy).  Synchronises the stream when it had work to do.
hg_status sort_large_sets(hg_ctx *c, const BatchPlan &pl, const uint32_t *h_cnt, size_t n, uint64_t threshold,
                          uint64_t *d_hits, const uint32_t *d_cnt, uint32_t *d_nd) {
  // keys per bucket aimed at (512-1 024 land in one; the sort's LDS is sized for four times that, hg_launch_sort_large) /
  // buckets per genome
  constexpr uint32_t TARGET = 1024, MAX_BUCKETS = 16384;
  std::vector<hg_bucket_job> jobs;
  std::vector<uint32_t> chunk_job, bucket_job, inplace;
  uint32_t cap_keys = 4 * TARGET;  // keys the bucket sort's LDS is sized for: four times what a bucket is expected to hold
  for (size_t g = 0; g < n; ++g) {
    const uint32_t cnt = std::min(h_cnt[g], pl.meta[g].hit_cap);
    if (cnt <= HG_SORT_LDS_MAX_KEYS) continue;
    uint32_t P = 2;
    while (P < MAX_BUCKETS && (uint64_t)P * TARGET < cnt) P <<= 1;
    // (the counting and scattering workgroups keep a genome's bucket counters in LDS up to PRIV buckets -- beyond that every
    // key pays a global atomic: a set of up to PRIV * 4 096 keys rather fills fewer, larger buckets)
    constexpr uint32_t PRIV = 2048;
    if (P > PRIV && (uint64_t)cnt <= (uint64_t)PRIV * 4096) P = PRIV;
    if (c->dbg_sort_buckets) {  // test hook (hg_ctx_set_debug): force overflowing buckets / the fallback
      P = (uint32_t)std::max(2, c->dbg_sort_buckets);
    } else if ((uint64_t)P * (HG_SORT_LDS_MAX_KEYS / 2) < cnt) {  // more than ~8 k keys per bucket expected: too many for LDS
      inplace.push_back((uint32_t)g);
      continue;
    }
    cap_keys = std::max<uint32_t>(cap_keys, 4 * ((cnt + P - 1) / P));  // (a genome with more keys than MAX_BUCKETS * TARGET fills its buckets further)
    hg_bucket_job j{};
    j.hit_off = pl.meta[g].hit_off, j.n = cnt, j.P = P, j.genome = (uint32_t)g;
    // bucket(h) = floor(h * P / threshold) for h < threshold, as a multiply-high by ceil(P * 2^64 / threshold)
    const unsigned __int128 num = ((unsigned __int128)P << 64) + threshold - 1;
    const unsigned __int128 q = num / (threshold ? threshold : 1);
    j.mul = q > (unsigned __int128)UINT64_MAX ? UINT64_MAX : (uint64_t)q;
    j.bucket_first = (uint32_t)bucket_job.size(), j.chunk_first = (uint32_t)chunk_job.size();
    bucket_job.insert(bucket_job.end(), P, (uint32_t)jobs.size());
    chunk_job.insert(chunk_job.end(), (cnt + HG_BUCKET_CHUNK - 1) / HG_BUCKET_CHUNK, (uint32_t)jobs.size());
    jobs.push_back(j);
  }
  if (jobs.empty() && inplace.empty()) return HG_OK;
  hg_status s;
  const size_t jb = (jobs.size() * sizeof(hg_bucket_job) + 63) & ~(size_t)63;
  const size_t cb = (chunk_job.size() * 4 + 63) & ~(size_t)63, bb = (bucket_job.size() * 4 + 63) & ~(size_t)63;
  const size_t kb = ((5 * bucket_job.size() + jobs.size()) * 4 + 63) & ~(size_t)63;
  const size_t tb = ((std::max(inplace.size(), jobs.size())) * 4 + 63) & ~(size_t)63;
  if ((s = hg_ensure(c, c->w_lsort, jb + cb + bb + kb + tb + 64)) != HG_OK) return s;
  auto *base = static_cast<uint8_t *>(c->w_lsort.p);
  auto *d_jobs = reinterpret_cast<hg_bucket_job *>(base);
  auto *d_chunk = reinterpret_cast<uint32_t *>(base + jb), *d_bucket = reinterpret_cast<uint32_t *>(base + jb + cb);
  auto *d_bk = reinterpret_cast<uint32_t *>(base + jb + cb + bb), *d_todo = reinterpret_cast<uint32_t *>(base + jb + cb + bb + kb);
  if (!jobs.empty()) {
    if ((s = hg_ensure(c, c->w_hits2, pl.total_slots * sizeof(uint64_t) + 16)) != HG_OK) return s;
    HG_HIP(c, hipMemcpyAsync(d_jobs, jobs.data(), jobs.size() * sizeof(hg_bucket_job), hipMemcpyHostToDevice, c->stream));
    HG_HIP(c, hipMemcpyAsync(d_chunk, chunk_job.data(), chunk_job.size() * 4, hipMemcpyHostToDevice, c->stream));
    HG_HIP(c, hipMemcpyAsync(d_bucket, bucket_job.data(), bucket_job.size() * 4, hipMemcpyHostToDevice, c->stream));
    std::vector<uint32_t> fail(jobs.size());
    {
      hg_timed tm(c, HG_T_SORT);
      HG_HIP(c, hg_launch_sort_large(c->stream, d_jobs, (uint32_t)jobs.size(), d_chunk, (uint32_t)chunk_job.size(), d_bucket,
                                     (uint32_t)bucket_job.size(), d_bk, d_hits, static_cast<uint64_t *>(c->w_hits2.p), d_nd,
                                     c->dbg_sort_buckets ? HG_SORT_LDS_MAX_KEYS : cap_keys));
    }
    HG_HIP(c, hipMemcpyAsync(fail.data(), d_bk + 5 * bucket_job.size(), jobs.size() * 4, hipMemcpyDeviceToHost, c->stream));
    HG_HIP(c, hipStreamSynchronize(c->stream));  // also keeps the host vectors alive until the uploads are done
    for (size_t k = 0; k < jobs.size(); ++k)
      if (fail[k]) inplace.push_back(jobs[k].genome);
  }
  if (!inplace.empty()) {
    HG_HIP(c, hipMemcpyAsync(d_todo, inplace.data(), inplace.size() * 4, hipMemcpyHostToDevice, c->stream));
    {
      hg_timed tm(c, HG_T_SORT);
      HG_HIP(c, hg_launch_sort_inplace(c->stream, static_cast<hg_genome_meta *>(c->w_gmeta.p), d_todo, (uint32_t)inplace.size(),
                                       d_hits, d_cnt, d_nd));
    }
    HG_HIP(c, hipStreamSynchronize(c->stream));
  }
  return HG_OK;
}

// Runs hash+sample and sort/unique.  On return (stream synchronised) the device hit buffer holds
// each genome's ascending distinct hashes at meta[g].hit_off and *d_ndistinct_out the counts.
// ASCII genomes -> hg_pack2 blobs on the device (stream-ordered): genome i of d_seq (seq_offs[i], lens[i]) to
// d_blobs + blob_offs[i].  The offset tables travel through the ctx's pinned scratch (overwritten: callers stage
// nothing there across this call).
hg_status pack_batch(hg_ctx *c, const uint8_t *d_seq, const uint64_t *seq_offs, const uint64_t *lens, size_t n,
                     uint32_t norm_mode, uint8_t *d_blobs, const uint64_t *blob_offs) {
  if (n == 0) return HG_OK;
  hg_status s;
  if ((s = hg_ensure(c, c->w_pktab, 3 * n * sizeof(uint64_t) + 64)) != HG_OK) return s;
  if ((s = hg_ensure_pinned(c, 3 * n * sizeof(uint64_t) + 64)) != HG_OK) return s;
  HG_HIP(c, hipStreamSynchronize(c->stream));  // (the scratch may still feed an earlier upload)
  auto *tab = static_cast<uint64_t *>(c->h_pin);
  uint64_t max_len = 0;
  for (size_t g = 0; g < n; ++g) {
    if ((seq_offs[g] & 3) || (blob_offs[g] & 15)) return hg_fail(c, HG_ERR_INVALID, "pack2: sequence offsets must be multiples of 4, blob offsets of 16");
    tab[3 * g] = seq_offs[g], tab[3 * g + 1] = lens[g], tab[3 * g + 2] = blob_offs[g];
    max_len = std::max(max_len, lens[g]);
  }
  HG_HIP(c, hipMemcpyAsync(c->w_pktab.p, tab, 3 * n * sizeof(uint64_t), hipMemcpyHostToDevice, c->stream));
  const uint64_t groups = ((((max_len + 7) / 8 + 15) & ~(uint64_t)15) + 3) / 4;  // lanes per genome: one per 4 bitmap bytes
  const uint64_t blocks = (groups + 255) / 256;
  if (blocks > 0x7FFFFFFFull) return hg_fail(c, HG_ERR_UNSUPPORTED, "pack2: genome too long for one launch");
  if (blocks)
    HG_HIP(c, hg_launch_pack2(c->stream, d_seq, static_cast<const uint64_t *>(c->w_pktab.p), (uint32_t)n, (uint32_t)blocks,
                              norm_mode == HG_NORM_U2T ? 1u : 0u, d_blobs));
  HG_HIP(c, hipStreamSynchronize(c->stream));  // (the pinned table is free again)
  return HG_OK;
}

// One-genome callers that want the sorted hash list on the host (hg_kmer_hash_sample): the distinct count and the first
// max_hashes hashes ride back with the counter copy sample_batch synchronises on anyway -- one synchronisation per call
// instead of three.  valid is set when the list the LDS sort produced is final (no overflow, no second sort pass).
struct SampleFetch {
  size_t max_hashes = 0;
  const uint64_t *h_hashes = nullptr;  // in the ctx's page-locked scratch: consume before the next call on the ctx
  uint32_t nd = 0;
  bool valid = false;
};

hg_status sample_batch(hg_ctx *c, const uint8_t *d_seq, const uint64_t *offsets, const uint64_t *lens,
                       size_t n, uint32_t ksize, uint64_t threshold, uint64_t scaled_for_cap, uint64_t seed,
                       bool canonical, uint32_t norm_mode, BatchPlan &pl, uint32_t **d_ndistinct_out, bool packed = false,
                       const uint64_t *mask_offs = nullptr, SampleFetch *fetch = nullptr) {
  if (n > 0x7FFFFFFFull) return hg_fail(c, HG_ERR_UNSUPPORTED, "more than 2^31 genomes in one batch");
  std::vector<uint64_t> hook_offs;
  if (!packed && c->dbg_kmer_input == "packed") {
    // test hook: the batch arrived as ASCII -- pack it here and run the packed kernels on the blobs, so that every
    // ASCII entry point (and with it every parity test) can be driven through both input forms
    hook_offs.resize(n);
    uint64_t total = 0;
    for (size_t g = 0; g < n; ++g) hook_offs[g] = total, total += hg_pack2_size(lens[g]);
    hg_status s;
    if ((s = hg_ensure(c, c->w_pk, total + 64)) != HG_OK) return s;
    if ((s = pack_batch(c, d_seq, offsets, lens, n, norm_mode, static_cast<uint8_t *>(c->w_pk.p), hook_offs.data())) != HG_OK) return s;
    d_seq = static_cast<const uint8_t *>(c->w_pk.p), offsets = hook_offs.data(), packed = true;
  }
  std::vector<uint32_t> want;
  for (int attempt = 0; attempt < 3; ++attempt) {
    hg_status s;
    // same geometry as the previous call (typical for a stream of equally shaped batches): the
    // work-item table and the per-genome records are still on the device
    const bool reuse = attempt == 0 && c->plan_valid && c->plan_ksize == ksize && c->plan_scaled == scaled_for_cap &&
                       c->plan_packed == packed && c->plan_offs.size() == n && std::memcmp(c->plan_offs.data(), offsets, n * 8) == 0 &&
                       std::memcmp(c->plan_lens.data(), lens, n * 8) == 0 &&
                       (mask_offs ? (c->plan_masks.size() == n && std::memcmp(c->plan_masks.data(), mask_offs, n * 8) == 0) : c->plan_masks.empty());
    size_t n_items;
    if (reuse) {
      n_items = c->plan_items;
      pl.total_slots = c->plan_slots, pl.max_cap = c->plan_max_cap;
      pl.meta.resize(n);
      uint64_t slot = 0;
      for (size_t g = 0; g < n; ++g) {  // only what callers read back: capacities and hit offsets
        pl.meta[g].hit_cap = c->plan_caps[g];
        pl.meta[g].hit_off = slot;
        slot += c->plan_caps[g];
      }
    } else {
      c->plan_valid = false;
      if ((s = make_plan(c, offsets, lens, n, ksize, scaled_for_cap, attempt ? &want : nullptr, pl, mask_offs)) != HG_OK) return s;
      n_items = pl.item_genome.size();
    }
    if ((s = hg_ensure(c, c->w_gmeta, n * sizeof(hg_genome_meta) + 16)) != HG_OK) return s;
    if ((s = hg_ensure(c, c->w_items, n_items * sizeof(uint32_t) + 16)) != HG_OK) return s;
    if ((s = hg_ensure(c, c->w_hits, pl.total_slots * sizeof(uint64_t) + 16)) != HG_OK) return s;
    if ((s = hg_ensure(c, c->w_cnt, 2 * n * sizeof(uint32_t) + 16)) != HG_OK) return s;
    const size_t pin_meta = (n * sizeof(hg_genome_meta) + 63) & ~(size_t)63;
    const size_t pin_items = (n_items * sizeof(uint32_t) + 63) & ~(size_t)63;
    // (the fetch block lies behind the plan's staging area whether this call uses that or not)
    const size_t fetch_off = ((n * sizeof(uint32_t) + 63) & ~(size_t)63) + pin_meta + pin_items;
    const size_t fetch_n = (fetch && n == 1) ? std::min<size_t>({fetch->max_hashes, pl.meta[0].hit_cap, (size_t)1 << 16}) : 0;
    if ((s = hg_ensure_pinned(c, n * sizeof(uint32_t) + 64 + (reuse ? 0 : pin_meta + pin_items) +
                                     (fetch && n == 1 ? fetch_off + 64 + fetch_n * 8 : 0))) != HG_OK) return s;
    auto *d_meta = static_cast<hg_genome_meta *>(c->w_gmeta.p);
    auto *d_items = static_cast<uint32_t *>(c->w_items.p);
    auto *d_hits = static_cast<uint64_t *>(c->w_hits.p);
    auto *d_cnt = static_cast<uint32_t *>(c->w_cnt.p);
    uint32_t *d_nd = d_cnt + n;
    auto *h_cnt = static_cast<uint32_t *>(c->h_pin);
    if (!reuse) {  // upload through pinned staging so that the copies are truly asynchronous
      uint8_t *pin = static_cast<uint8_t *>(c->h_pin) + ((n * sizeof(uint32_t) + 63) & ~(size_t)63);
      std::memcpy(pin, pl.meta.data(), n * sizeof(hg_genome_meta));
      HG_HIP(c, hipMemcpyAsync(d_meta, pin, n * sizeof(hg_genome_meta), hipMemcpyHostToDevice, c->stream));
      if (n_items) {
        std::memcpy(pin + pin_meta, pl.item_genome.data(), n_items * sizeof(uint32_t));
        HG_HIP(c, hipMemcpyAsync(d_items, pin + pin_meta, n_items * sizeof(uint32_t), hipMemcpyHostToDevice, c->stream));
      }
    }
    HG_HIP(c, hipMemsetAsync(d_cnt, 0, 2 * n * sizeof(uint32_t), c->stream));
    {
      hg_timed tm(c, HG_T_KMER);
      c->last_kernel[HG_T_KMER] = hg_kmer_kernel_name(ksize, canonical, packed);
      HG_HIP(c, hg_launch_kmer_sample(c->stream, d_seq, d_meta, d_items, (uint32_t)n_items, ksize, threshold,
                                      seed, canonical, norm_mode, d_hits, d_cnt, packed));
    }
    uint32_t sort_cap = pl.max_cap;
    {
      // The LDS sort is sized by the genomes' CAPACITIES (twice the expected count + 1 024: 64 KiB for a 5 Mbp genome,
      // two workgroups per CU).  When the plan is a repeat, the counts of its last run are known: size by those (+ 12.5 %,
      // 32 KiB -> five workgroups per CU); a genome that outgrows it is left to the large-set path below, as always.
      if (reuse && c->plan_max_hits) sort_cap = (uint32_t)std::min<uint64_t>(sort_cap, (uint64_t)c->plan_max_hits + c->plan_max_hits / 8 + 16);
      hg_timed tm(c, HG_T_SORT, HG_T_KMER);
      HG_HIP(c, hg_launch_sort_unique(c->stream, d_meta, (uint32_t)n, d_hits, d_cnt, d_nd, sort_cap, threshold));
    }
    // overflow check on the raw counters (they keep counting past the capacity)
    HG_HIP(c, hipMemcpyAsync(h_cnt, d_cnt, n * sizeof(uint32_t), hipMemcpyDeviceToHost, c->stream));
    uint8_t *h_fetch = static_cast<uint8_t *>(c->h_pin) + fetch_off;
    if (fetch && n == 1) {
      fetch->valid = false;
      HG_HIP(c, hipMemcpyAsync(h_fetch, d_nd, sizeof(uint32_t), hipMemcpyDeviceToHost, c->stream));
      if (fetch_n)
        HG_HIP(c, hipMemcpyAsync(h_fetch + 64, d_hits + pl.meta[0].hit_off, fetch_n * sizeof(uint64_t), hipMemcpyDeviceToHost, c->stream));
    }
    HG_HIP(c, hipStreamSynchronize(c->stream));
    bool overflow = false;
    want.assign(n, 0);
    for (size_t g = 0; g < n; ++g)
      if (h_cnt[g] > pl.meta[g].hit_cap) overflow = true, want[g] = h_cnt[g];
    if (!overflow) {
      if (hg_sort_lds_keys(sort_cap) < hg_sort_lds_keys(pl.max_cap)) {
        // the count-sized sort left out every genome that grew past its size: those again, with the capacity-sized one
        std::vector<uint32_t> redo;
        const uint32_t keys = hg_sort_lds_keys(sort_cap);
        for (size_t g = 0; g < n; ++g) {
          const uint32_t cnt = std::min(h_cnt[g], pl.meta[g].hit_cap);
          if (cnt > keys && cnt <= HG_SORT_LDS_MAX_KEYS) redo.push_back((uint32_t)g);
        }
        if (!redo.empty()) {
          if ((s = hg_ensure(c, c->w_redo, redo.size() * 4 + 64)) != HG_OK) return s;
          // staged in the ctx's page-locked scratch behind the counters (which were consumed above) and uploaded on the
          // ctx's own stream like every other command of this path: no legacy-stream copy that would also synchronise
          // with the other ctxs of the device
          const size_t redo_off = (n * sizeof(uint32_t) + 63) & ~(size_t)63;
          if (c->h_pin_cap < redo_off + redo.size() * 4) {
            std::vector<uint32_t> keep(h_cnt, h_cnt + n);  // (growing the scratch frees the block the counters live in)
            if ((s = hg_ensure_pinned(c, redo_off + redo.size() * 4)) != HG_OK) return s;
            h_cnt = static_cast<uint32_t *>(c->h_pin);
            std::memcpy(h_cnt, keep.data(), n * sizeof(uint32_t));
          }
          uint32_t *h_redo = reinterpret_cast<uint32_t *>(static_cast<uint8_t *>(c->h_pin) + redo_off);
          std::memcpy(h_redo, redo.data(), redo.size() * 4);
          HG_HIP(c, hipMemcpyAsync(c->w_redo.p, h_redo, redo.size() * 4, hipMemcpyHostToDevice, c->stream));
          HG_HIP(c, hipStreamSynchronize(c->stream));  // (rare path; the next call may rewrite the scratch at once)
          hg_timed tm(c, HG_T_SORT);
          HG_HIP(c, hg_launch_sort_unique_todo(c->stream, d_meta, static_cast<uint32_t *>(c->w_redo.p), (uint32_t)redo.size(),
                                               d_hits, d_cnt, d_nd, pl.max_cap, threshold));
        }
      }
      // hash sets beyond the LDS sort: bucketed multi-workgroup sort (or, where that cannot work, in place)
      if ((s = sort_large_sets(c, pl, h_cnt, n, threshold, d_hits, d_cnt, d_nd)) != HG_OK) return s;
      h_cnt = static_cast<uint32_t *>(c->h_pin);  // (unchanged unless the pinned scratch grew)
      pl.big.clear();
      pl.max_hits = 0;
      for (size_t g = 0; g < n; ++g) {
        const uint32_t cnt = std::min(h_cnt[g], pl.meta[g].hit_cap);
        pl.max_hits = std::max(pl.max_hits, cnt);
        if (cnt > HG_ENC_SLAB) pl.big.emplace_back((uint32_t)g, cnt);
      }
      if (!reuse) {  // remember this plan for the next call
        c->plan_offs.assign(offsets, offsets + n);
        c->plan_lens.assign(lens, lens + n);
        if (mask_offs) c->plan_masks.assign(mask_offs, mask_offs + n);
        else c->plan_masks.clear();
        c->plan_caps.resize(n);
        for (size_t g = 0; g < n; ++g) c->plan_caps[g] = pl.meta[g].hit_cap;
        c->plan_ksize = ksize, c->plan_scaled = scaled_for_cap, c->plan_packed = packed;
        c->plan_slots = pl.total_slots, c->plan_max_cap = pl.max_cap, c->plan_items = n_items;
        c->plan_valid = true;
      }
      c->plan_max_hits = pl.max_hits;
      *d_ndistinct_out = d_nd;
      if (fetch && n == 1) {
        // the copies above saw the final list iff the first sort pass covered the set
        const uint32_t cnt = std::min(h_cnt[0], pl.meta[0].hit_cap);
        if (cnt <= hg_sort_lds_keys(sort_cap) && cnt <= HG_SORT_LDS_MAX_KEYS) {  // (then nothing above touched the scratch either)
          uint32_t nd;
          std::memcpy(&nd, h_fetch, sizeof nd);
          if (nd <= fetch_n) fetch->valid = true, fetch->nd = nd, fetch->h_hashes = reinterpret_cast<const uint64_t *>(h_fetch + 64);
        }
      }
      return HG_OK;
    }
    c->plan_valid = false;
  }
  return hg_fail(c, HG_ERR_HIP, "hit buffer overflow persisted after resizing");
}

hg_status check_params(hg_ctx *c, const hg_sketch_params *p) {
  if (!p) return hg_fail(c, HG_ERR_INVALID, "params == NULL");
  if (p->ksize < 1) return hg_fail(c, HG_ERR_INVALID, "ksize must be >= 1");
  if (p->ksize > 255) return hg_fail(c, HG_ERR_UNSUPPORTED, "ksize must be <= 255 (the reference's -k is u8)");
  if (p->scaled < 1) return hg_fail(c, HG_ERR_INVALID, "scaled must be >= 1");
  if (p->hv_layout > HG_LAYOUT_AVX2 || p->norm_mode > HG_NORM_U2T) return hg_fail(c, HG_ERR_INVALID, "bad layout / norm mode");
  if (p->hv_d == 0 || p->hv_d > 32768) return hg_fail(c, HG_ERR_UNSUPPORTED, "hv_d must be in 1..32768");
  return HG_OK;
}

}  // namespace

static hg_status sketch_batch_dev_impl(hg_ctx *c, const uint8_t *d_seq, const uint64_t *offsets, const uint64_t *lens, size_t n,
                                       const hg_sketch_params *p, int16_t *d_hv, int32_t *d_norm2, uint32_t *d_nhash, bool packed,
                                       const uint64_t *mask_offs = nullptr);

hg_status hg_sketch_batch_dev_packed_masks(hg_ctx *c, const uint8_t *d_blobs, const uint64_t *code_offs, const uint64_t *mask_offs,
                                           const uint64_t *n_bps, size_t n, const hg_sketch_params *p, int16_t *d_hv,
                                           int32_t *d_norm2, uint32_t *d_nhash) {
  return sketch_batch_dev_impl(c, d_blobs, code_offs, n_bps, n, p, d_hv, d_norm2, d_nhash, true, mask_offs);
}

extern "C" hg_status hg_sketch_batch_dev(hg_ctx *c, const uint8_t *d_seq, const uint64_t *offsets,
                                         const uint64_t *lens, size_t n, const hg_sketch_params *p,
                                         int16_t *d_hv, int32_t *d_norm2, uint32_t *d_nhash) {
  return sketch_batch_dev_impl(c, d_seq, offsets, lens, n, p, d_hv, d_norm2, d_nhash, false);
}

extern "C" hg_status hg_sketch_batch_dev_packed(hg_ctx *c, const uint8_t *d_blobs, const uint64_t *offsets,
                                                const uint64_t *n_bps, size_t n, const hg_sketch_params *p,
                                                int16_t *d_hv, int32_t *d_norm2, uint32_t *d_nhash) {
  if (c && offsets)
    for (size_t g = 0; g < n; ++g)
      if (offsets[g] & 15) return hg_fail(c, HG_ERR_INVALID, "blob offsets must be multiples of 16");
  return sketch_batch_dev_impl(c, d_blobs, offsets, n_bps, n, p, d_hv, d_norm2, d_nhash, true);
}

extern "C" hg_status hg_pack2_batch_dev(hg_ctx *c, const uint8_t *d_seq, const uint64_t *offsets, const uint64_t *lens, size_t n,
                                        uint32_t norm_mode, uint8_t *d_blobs, const uint64_t *blob_offsets) {
  if (!c) return HG_ERR_INVALID;
  if (n == 0) return HG_OK;
  if (!d_seq || !offsets || !lens || !d_blobs || !blob_offsets || norm_mode > HG_NORM_U2T) return hg_fail(c, HG_ERR_INVALID, "bad argument");
  if (n > 0x7FFFFFFFull) return hg_fail(c, HG_ERR_UNSUPPORTED, "more than 2^31 genomes in one batch");
  HG_HIP(c, hipSetDevice(c->device));
  return pack_batch(c, d_seq, offsets, lens, n, norm_mode, d_blobs, blob_offsets);
}

extern "C" hg_status hg_pack2_dev(hg_ctx *c, const uint8_t *d_seq, size_t n_bps, uint32_t norm_mode, uint8_t *d_blob) {
  if (!c) return HG_ERR_INVALID;
  if (((uintptr_t)d_seq & 3) || ((uintptr_t)d_blob & 15)) return hg_fail(c, HG_ERR_INVALID, "hg_pack2_dev: d_seq must be 4-byte, d_blob 16-byte aligned");
  const uint64_t zero = 0, len = n_bps;
  return hg_pack2_batch_dev(c, d_seq, &zero, &len, 1, norm_mode, d_blob, &zero);
}

static hg_status sketch_batch_dev_impl(hg_ctx *c, const uint8_t *d_seq, const uint64_t *offsets, const uint64_t *lens, size_t n,
                                       const hg_sketch_params *p, int16_t *d_hv, int32_t *d_norm2, uint32_t *d_nhash, bool packed,
                                       const uint64_t *mask_offs) {
  if (!c) return HG_ERR_INVALID;
  hg_status s = check_params(c, p);
  if (s != HG_OK) return s;
  if (n == 0) return HG_OK;
  if (!d_seq || !offsets || !lens || !d_hv || !d_norm2 || !d_nhash) return hg_fail(c, HG_ERR_INVALID, "NULL argument");
  HG_HIP(c, hipSetDevice(c->device));
  BatchPlan pl;
  uint32_t *d_nd = nullptr;
  const uint64_t threshold = UINT64_MAX / p->scaled;  // src/sketch.rs:73
  s = sample_batch(c, d_seq, offsets, lens, n, p->ksize, threshold, p->scaled, p->seed, p->canonical != 0,
                   p->norm_mode, pl, &d_nd, packed, mask_offs);
  if (s != HG_OK) return s;
  // genomes with very large hash sets are encoded by several workgroups each (plan from the raw hit counts)
  hg_encode_split split{};
  std::vector<uint32_t> items, genomes;
  if (!pl.big.empty() && pl.big.size() < 65536) {
    for (size_t k = 0; k < pl.big.size(); ++k) {
      const uint32_t slabs = std::min<uint32_t>((pl.big[k].second + HG_ENC_SLAB - 1) / HG_ENC_SLAB, 65535u);
      for (uint32_t sl = 0; sl < slabs; ++sl) items.push_back(pl.big[k].first), items.push_back(sl | ((uint32_t)k << 16));
      genomes.push_back(pl.big[k].first);
    }
    const size_t ib = (items.size() * 4 + 63) & ~(size_t)63, gb = (genomes.size() * 4 + 63) & ~(size_t)63;
    if ((s = hg_ensure(c, c->w_lsort, ib + gb + 64)) != HG_OK) return s;
    if ((s = hg_ensure(c, c->w_hits2, genomes.size() * (size_t)p->hv_d * 4 + 64)) != HG_OK) return s;
    auto *d_items = static_cast<uint32_t *>(c->w_lsort.p);
    auto *d_genomes = reinterpret_cast<uint32_t *>(static_cast<uint8_t *>(c->w_lsort.p) + ib);
    HG_HIP(c, hipMemcpyAsync(d_items, items.data(), items.size() * 4, hipMemcpyHostToDevice, c->stream));
    HG_HIP(c, hipMemcpyAsync(d_genomes, genomes.data(), genomes.size() * 4, hipMemcpyHostToDevice, c->stream));
    split.d_items = d_items, split.d_genomes = d_genomes, split.d_accum = static_cast<uint32_t *>(c->w_hits2.p);
    split.n_items = (uint32_t)(items.size() / 2), split.n_genomes = (uint32_t)genomes.size();
  }
  {
    hg_timed tm(c, HG_T_ENCODE);
    HG_HIP(c, hg_launch_encode(c->stream, static_cast<hg_genome_meta *>(c->w_gmeta.p), (uint32_t)n,
                               static_cast<uint64_t *>(c->w_hits.p), d_nd, p->hv_d, p->hv_layout, d_hv, d_norm2,
                               split.n_items ? &split : nullptr, pl.max_hits));
  }
  if (split.n_items) HG_HIP(c, hipStreamSynchronize(c->stream));  // the pageable item tables must outlive their upload
  HG_HIP(c, hipMemcpyAsync(d_nhash, d_nd, n * sizeof(uint32_t), hipMemcpyDeviceToDevice, c->stream));
  return HG_OK;
}

// NUMA node the device hangs off (sysfs of its PCI function), -1 when unknown.  Page-locked buffers filled by
// threads of that node are fetched ~25 % faster than buffers on the other socket (2-socket EPYC host, measured).
extern "C" int hg_device_numa_node(int device_id) {
  char bus[64] = {0};
  if (hipDeviceGetPCIBusId(bus, (int)sizeof bus, device_id) != hipSuccess) return -1;
  for (char *q = bus; *q; ++q) *q = (char)std::tolower((unsigned char)*q);
  const std::string path = std::string("/sys/bus/pci/devices/") + bus + "/numa_node";
  FILE *f = std::fopen(path.c_str(), "r");
  if (!f) return -1;
  int node = -1;
  if (std::fscanf(f, "%d", &node) != 1) node = -1;
  std::fclose(f);
  return node;
}

// ---- page-locked read buffers -----------------------------------------------------------------------------------
// A pageable hipMemcpyAsync goes through the runtime's bounce buffer and blocks its caller; sequence read straight
// into page-locked memory is DMA'd by hg_sketch_batch at the link rate instead.
namespace {
bool grow_pinned(uint8_t *&buf, size_t &cap, size_t need, size_t keep, void *) {
  if (need <= cap) return true;
  // recycled slots see files of similar but not equal sizes: round up so that they rarely move
  const size_t want = ((need + need / 8) + ((size_t)1 << 20) - 1) & ~(((size_t)1 << 20) - 1);
  void *nb = nullptr;
  if (hipHostMalloc(&nb, want, hipHostMallocPortable) != hipSuccess || !nb) return false;
  if (buf && keep) std::memcpy(nb, buf, std::min(keep, cap));
  if (buf) (void)hipHostFree(buf);
  buf = static_cast<uint8_t *>(nb), cap = want;
  return true;
}
}  // namespace

extern "C" hg_status hg_read_fastx_pinned(const char *path, uint32_t mode, uint8_t **buf, size_t *cap, size_t *n_bps) {
  return hg_read_fastx_impl(path, mode, buf, cap, n_bps, grow_pinned, nullptr);
}

extern "C" void hg_pinned_free(void *p) {
  if (p) (void)hipHostFree(p);
}

// Host-fed batch.  The batch is cut into sub-batches of about HG_STAGE_BYTES; a helper thread queues their
// uploads on the context's copy stream (one event per sub-batch) while this thread runs hash/sort/encode
// of the sub-batches already on the device, so PCIe transfer and kernels overlap for pinned and for
// pageable caller memory alike (a pageable hipMemcpyAsync blocks the thread that issues it).
constexpr uint64_t HG_STAGE_BYTES = 64ull << 20;
constexpr uint64_t HG_PACK_BYTES = HG_STAGE_BYTES + (2ull << 20);  // a sub-batch of genomes < 1 MiB each fits

// Host threads the library may use for its own host-side work on a call (2-bit packing of a host-fed batch): the cores
// this process may run on, at most 16 -- the reference's default `-t` (src/utils.rs:54-56).
static unsigned host_threads() {
  unsigned n = std::thread::hardware_concurrency();
  cpu_set_t set;
  if (sched_getaffinity(0, sizeof set, &set) == 0) n = std::min<unsigned>(n ? n : 1u, (unsigned)CPU_COUNT(&set));
  return std::max(1u, std::min(n, 16u));
}

// Host-fed calls in flight in this process (hg_sketch_batch / hg_kmer_hash_sample, any context): the reference's pattern is
// one call per genome from a pool of host threads (src/sketch_cuda.rs:79-96), and then the calls share ONE link.
static std::atomic<int> g_hostfed_calls{0};
namespace {
struct HostfedCall {
  int others;
  HostfedCall() : others(g_hostfed_calls.fetch_add(1)) {}
  ~HostfedCall() { g_hostfed_calls.fetch_sub(1); }
};
}  // namespace
static bool host_pinned(const void *p) {
  hipPointerAttribute_t a;
  if (hipPointerGetAttributes(&a, p) != hipSuccess) {
    (void)hipGetLastError();  // (an ordinary malloc'ed pointer is "invalid value" to the runtime)
    return false;
  }
  return a.type == hipMemoryTypeHost;
}
// Whether ONE genome handed over by a host-fed call goes over the link 2-bit packed (by the calling thread into the
// context's page-locked staging buffer) instead of as ASCII.
//  * pageable source (>= 256 KB): packed -- the runtime would stage it through its own pinned buffers anyway;
//  * page-locked source, fewer than 3 other host-fed calls in flight: ASCII (a lone 5 Mbp call takes 0.18 ms as ASCII,
//    0.33 ms packed);
//  * page-locked source, the link shared by K >= 4 calls (the reference's one-call-per-genome pattern from a thread
//    pool): packed as long as the host keeps up.  With K calls sharing a link of L bytes/s a call waits n K / L for its
//    ASCII, or n / r + 0.375 n K / L packed at r bytes/s: packing pays while r > L / (0.625 K).  r is what the calling
//    threads really achieve TOGETHER (16 of them are bound by host DRAM: 7 GB/s each on a quiet box of the pool -- 17 k
//    genomes/s against 10 k --, 4 GB/s on one whose memory was busy -- 8.6 k against 10 k), so it is measured in the
//    packed calls themselves (decayed mean, kept per range of K: r falls as K grows); when it falls short, the next calls
//    of that range go as ASCII -- 256 of them, doubling each time packing fails again, up to 16 384 -- and then packing is
//    tried afresh.
// Hook: "hostfed" = "ascii" never, "packed" always.
namespace {
constexpr double HG_LINK_BYTES_PER_S = 50e9;  // what ASCII uploads from page-locked memory reach on Gen5 x16 (bench.py host_fed.ascii_link)
struct PackState {
  std::atomic<uint64_t> rate{0};       // decayed mean of the bytes/s one calling thread packed at, contended packed calls
  std::atomic<uint32_t> samples{0};    // ... and how many calls it has seen since packing was (re)started
  std::atomic<int32_t> ascii_left{0};  // > 0: contended calls still to go as ASCII before packing is tried again
  std::atomic<uint32_t> backoff{256};
};
PackState g_pack[6];  // by calls in flight: 4-5, 6-7, 8-11, 12-15, 16-23, 24 and more
inline PackState &pack_state(int sharing) {
  return g_pack[sharing < 6 ? 0 : sharing < 8 ? 1 : sharing < 12 ? 2 : sharing < 16 ? 3 : sharing < 24 ? 4 : 5];
}
struct SingleChoice {
  bool packed = false, measured = false;
  int sharing = 1;  // calls in flight, this one included
};
// after a measured call packed its genome: n bytes in sec seconds
void pack_measured(const SingleChoice &ch, uint64_t n, double sec) {
  if (!ch.measured || sec <= 0) return;
  PackState &st = pack_state(ch.sharing);
  const uint64_t rate = (uint64_t)((double)n / sec), old = st.rate.load(std::memory_order_relaxed);
  const uint64_t now = old ? old - old / 8 + rate / 8 : rate;  // (racing updates lose a sample at worst)
  st.rate.store(now, std::memory_order_relaxed);
  const uint32_t seen = st.samples.fetch_add(1, std::memory_order_relaxed) + 1;
  if (seen >= 8 && (double)now * 0.625 * ch.sharing < 0.9 * HG_LINK_BYTES_PER_S) {
    const uint32_t b = st.backoff.load(std::memory_order_relaxed);
    st.ascii_left.store((int32_t)b, std::memory_order_relaxed);
    st.backoff.store(std::min<uint32_t>(2 * b, 16384u), std::memory_order_relaxed);
    st.samples.store(0, std::memory_order_relaxed), st.rate.store(0, std::memory_order_relaxed);
  } else if (seen == 1024) {
    st.backoff.store(256, std::memory_order_relaxed);  // (a long run of packing that paid)
  }
}
}  // namespace
static SingleChoice pack_single(const hg_ctx *c, const void *seq, uint64_t n_bps, int others) {
  SingleChoice r;
  r.sharing = others + 1;
  if (c->dbg_hostfed == "ascii" || hg_pack2_size(n_bps) > HG_PACK_BYTES) return r;
  r.packed = true;
  if (c->dbg_hostfed == "packed") return r;
  r.packed = false;
  if (n_bps < (256u << 10)) return r;
  if (!host_pinned(seq)) {
    r.packed = true;
    return r;
  }
  if (others < 3) return r;
  PackState &st = pack_state(r.sharing);
  if (st.ascii_left.load(std::memory_order_relaxed) > 0) {
    st.ascii_left.fetch_sub(1, std::memory_order_relaxed);
    return r;
  }
  r.packed = r.measured = true;
  return r;
}

// Page-locked staging buffer b of the context with room for `need` bytes, free to be rewritten (the upload that last read
// it has passed).  Sized by need -- locking pages costs ~0.25 ms per MB, and a pool of one-call-per-genome contexts
// should not pin 66 MB each for 2 MB blobs.
static hipError_t pack_buf_for(hg_ctx *c, int b, size_t need) {
  hipError_t e = hipSuccess;
  if (c->pack_used[b]) e = hipEventSynchronize(c->pack_ev[b]);
  if (e != hipSuccess) return e;
  if (!c->pack_ev[b] && (e = hipEventCreateWithFlags(&c->pack_ev[b], hipEventDisableTiming)) != hipSuccess) return e;
  if (c->pack_cap[b] >= need) return hipSuccess;
  if (c->pack_buf[b]) (void)hipHostFree(c->pack_buf[b]);
  c->pack_buf[b] = nullptr, c->pack_cap[b] = 0, c->pack_used[b] = false;
  const size_t want = (need + need / 4 + ((size_t)1 << 20) - 1) & ~(((size_t)1 << 20) - 1);
  if ((e = hipHostMalloc(&c->pack_buf[b], want, hipHostMallocDefault)) != hipSuccess) return e;
  c->pack_cap[b] = want;
  return hipSuccess;
}

extern "C" hg_status hg_sketch_batch(hg_ctx *c, const uint8_t *const *seqs, const size_t *lens, size_t n,
                                     const hg_sketch_params *p, int16_t *hv_out, int32_t *norm2_out,
                                     uint32_t *nhash_out) {
  if (!c) return HG_ERR_INVALID;
  hg_status s = check_params(c, p);
  if (s != HG_OK) return s;
  if (n == 0) return HG_OK;
  if (!seqs || !lens || !hv_out || !norm2_out || !nhash_out) return hg_fail(c, HG_ERR_INVALID, "NULL argument");
  HG_HIP(c, hipSetDevice(c->device));
  // The link is what limits this entry point (50 GB/s = 10 k genomes/s of 5 Mbp as ASCII): a batch that is worth it goes
  // over as 2-bit packed bases -- hg_pack2 blobs, 0.375 bytes per base, packed by a few host threads of this call into the
  // page-locked staging buffers while the previous sub-batch uploads -- and is sketched by the packed-input kernels
  // (bit-identical results).  Needs cores: with fewer than 4 usable ones the ASCII path stays.  Hook: "hostfed" = "ascii".
  // A call that hands over little (the n = 1 of the one-call-per-genome pattern) packs on its own thread, when
  // pack_single() says the link is contended.
  HostfedCall in_flight;
  unsigned P = host_threads();
  uint64_t all_bytes = 0;
  for (size_t g = 0; g < n; ++g) all_bytes += lens[g];
  // (batches of genomes below 1 kbp on average stay ASCII: hg_pack2 blobs carry 32 bytes of padding each)
  bool want_pack = (P >= 4 && all_bytes >= (32ull << 20) && all_bytes / n >= (1u << 10) && c->dbg_hostfed != "ascii") ||
                   (n > 1 && c->dbg_hostfed == "packed");
  if (want_pack) P = std::max(1u, P / (unsigned)(1 + in_flight.others));
  SingleChoice single;
  if (!want_pack && n == 1) {
    single = pack_single(c, seqs[0], lens[0], in_flight.others);
    if (single.packed) want_pack = true, P = 1;
  }
  const uint64_t stage_bytes = want_pack ? 2 * HG_STAGE_BYTES : HG_STAGE_BYTES;  // (packed: 48 MB per upload)
  // device layout: 16-byte aligned starts, 64 bytes of slack; sub-batch boundaries by bytes
  std::vector<uint64_t> offs(n), l64(n), boffs(n);
  std::vector<size_t> cut{0};
  uint64_t total = 0, in_chunk = 0;
  for (size_t g = 0; g < n; ++g) {
    if (lens[g] && !seqs[g]) return hg_fail(c, HG_ERR_INVALID, "NULL sequence");
    if (in_chunk >= stage_bytes) cut.push_back(g), in_chunk = 0;
    offs[g] = total, l64[g] = lens[g];
    const uint64_t padded = (lens[g] + 15) & ~(uint64_t)15;
    total += padded, in_chunk += padded;
  }
  cut.push_back(n);
  const size_t n_chunks = cut.size() - 1;
  // packed sub-batches: blob g of sub-batch k at the sub-batch's own start in the device buffer (its ASCII region is
  // larger than its blobs) + the sum of the blob sizes in front of it; a sub-batch whose blobs outgrow a staging buffer
  // (one huge genome) stays ASCII
  std::vector<uint8_t> sub_packed(n_chunks, 0);
  std::vector<uint64_t> sub_pk_bytes(n_chunks, 0);
  if (want_pack) {
    for (size_t k = 0; k < n_chunks; ++k) {
      uint64_t at = 0;
      for (size_t g = cut[k]; g < cut[k + 1]; ++g) boffs[g] = offs[cut[k]] + at, at += hg_pack2_size(lens[g]);
      sub_pk_bytes[k] = at;
      // ... and one whose blobs outgrow its own ASCII region (hg_pack2_size is 32 for 1..16 bases, their padded ASCII 16:
      // a sub-batch of very short sequences) stays ASCII too -- its blobs would run into the next sub-batch's region, or,
      // for the last one, past the end of the buffer
      const uint64_t span = offs[cut[k + 1] - 1] + ((lens[cut[k + 1] - 1] + 15) & ~(uint64_t)15) - offs[cut[k]];
      sub_packed[k] = at <= HG_PACK_BYTES && at <= span && at > 0;
    }
  }
  std::unique_ptr<CallPool> pool;
  // The pool's threads run on the NUMA node the sequences lie on (page-locked memory from the HIP runtime: the device's node,
  // like the staging buffers they write; packing from the other socket is ~1.5x slower), the uploader thread too.
  int pack_node = -1;
  if (want_pack && P > 1) {
    for (size_t g = 0; g < n && pack_node < 0; ++g)
      if (lens[g]) pack_node = hg_numa_node_of(seqs[g]);
    if (pack_node < 0) pack_node = hg_device_numa_node(c->device);
  }
  if (want_pack) pool.reset(new CallPool(P, pack_node));  // (takes the threads it can get)
  // packing has to outrun the link to be worth it from page-locked sources (ASCII goes at ~50 GB/s from those): the
  // uploader times its first packed sub-batch and leaves the rest as ASCII when the host is too slow for that
  const bool src_pinned = want_pack && n > 1 && host_pinned(seqs[0]);
  if ((s = hg_ensure(c, c->w_seq, total + 64)) != HG_OK) return s;
  const size_t hv_bytes = n * (size_t)p->hv_d * sizeof(int16_t);
  if ((s = hg_ensure(c, c->w_hv, hv_bytes + n * 8 + 64)) != HG_OK) return s;
  auto *d_seq = static_cast<uint8_t *>(c->w_seq.p);
  auto *d_hv = static_cast<int16_t *>(c->w_hv.p);
  auto *d_n2 = reinterpret_cast<int32_t *>(static_cast<uint8_t *>(c->w_hv.p) + ((hv_bytes + 15) & ~(size_t)15));
  auto *d_nh = reinterpret_cast<uint32_t *>(d_n2 + n);
  // one sub-batch (the n = 1 of a one-call-per-genome pool above all): its upload goes on the context's own stream, in
  // front of its kernels -- no second stream, no event to wait for
  const bool one_stream = n_chunks == 1;
  if (!one_stream && !c->copy_stream) HG_HIP(c, hipStreamCreateWithFlags(&c->copy_stream, hipStreamNonBlocking));
  const hipStream_t up_stream = one_stream ? c->stream : c->copy_stream;
  while (!one_stream && c->copy_events.size() < n_chunks) {
    hipEvent_t e;
    HG_HIP(c, hipEventCreateWithFlags(&e, hipEventDisableTiming));
    c->copy_events.push_back(e);
  }

  std::mutex mu;
  std::condition_variable cv;
  size_t queued = 0;  // sub-batches whose uploads and event are queued
  hipError_t copy_err = hipSuccess;
  auto upload = [&](size_t first_chunk) {
    hipError_t e = hipSetDevice(c->device);
    if (n_chunks > 1) (void)hg_bind_thread_to_numa_node(pack_node, P);  // (the helper thread only, never the caller's)
    for (size_t k = first_chunk; k < n_chunks; ++k) {
      const size_t g0 = cut[k], g1 = cut[k + 1];
      const uint64_t span = offs[g1 - 1] + ((l64[g1 - 1] + 15) & ~(uint64_t)15) - offs[g0];
      if (sub_packed[k]) {
        // 2-bit pack the sub-batch into page-locked staging (all host threads of the call), then ONE upload
        const int b = (int)(k & 1);
        if (e == hipSuccess) e = pack_buf_for(c, b, n == 1 ? sub_pk_bytes[k] : HG_PACK_BYTES);
        if (e == hipSuccess) {
          auto *pin = static_cast<uint8_t *>(c->pack_buf[b]);
          // pieces of 1 Mbase, so that the threads finish together whatever the genome sizes
          constexpr uint64_t PIECE = 1ull << 20;
          std::vector<std::pair<size_t, uint64_t>> pieces;
          for (size_t g = g0; g < g1; ++g)
            for (uint64_t b = 0; b < lens[g]; b += PIECE) pieces.emplace_back(g, b);
          // ... handed out in runs of at least 256 kbase: a task per 5 kbp genome cost more in the pool's hand-overs than
          // in packing (100 000 x 5 kbp: 107 ms packed against 30 ms as ASCII through one staging copy)
          std::vector<size_t> task_first{0};
          {
            uint64_t in_task = 0;
            for (size_t i = 0; i < pieces.size(); ++i) {
              if (in_task >= (256u << 10)) task_first.push_back(i), in_task = 0;
              in_task += std::min<uint64_t>(lens[pieces[i].first] - pieces[i].second, PIECE);
            }
            task_first.push_back(pieces.size());
          }
          const auto t0 = std::chrono::steady_clock::now();
          pool->run(task_first.size() - 1, [&](size_t t) {
            for (size_t i = task_first[t]; i < task_first[t + 1]; ++i) {
              const size_t g = pieces[i].first;
              const uint64_t b = pieces[i].second;
              hg_pack2_piece(seqs[g], lens[g], p->norm_mode, pin + (boffs[g] - boffs[g0]), b, std::min<uint64_t>(lens[g], b + PIECE));
            }
          });
          const double sec = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
          // (page-locked sources go at ~55 GB/s as ASCII; pageable ones through the runtime's bounce buffer at ~12 GB/s: a
          // host whose cores are capped by a cgroup quota -- host_threads() cannot see one -- may pack slower than even that)
          if (n > 1 && k == first_chunk && (double)span < (src_pinned ? 55e9 : 12e9) * sec && c->dbg_hostfed != "packed")
            for (size_t j = k + 1; j < n_chunks; ++j) sub_packed[j] = 0;
          if (n == 1) pack_measured(single, lens[0], sec);
          if (e == hipSuccess) e = hipMemcpyAsync(d_seq + boffs[g0], pin, sub_pk_bytes[k], hipMemcpyHostToDevice, up_stream);
          if (e == hipSuccess) e = hipEventRecord(c->pack_ev[b], up_stream);
          c->pack_used[b] = true;
        }
      } else if (g1 - g0 >= 16 && span / (g1 - g0) < ((uint64_t)1 << 20) && span <= HG_PACK_BYTES) {
        // many small genomes: pack them into pinned memory (device layout) and upload once -- a
        // hipMemcpyAsync per 2 kbp genome costs more than the genome
        const int b = (int)(k & 1);
        if (e == hipSuccess) e = pack_buf_for(c, b, HG_PACK_BYTES);
        if (e == hipSuccess) {
          auto *pin = static_cast<uint8_t *>(c->pack_buf[b]);
          for (size_t g = g0; g < g1; ++g)
            if (lens[g]) std::memcpy(pin + (offs[g] - offs[g0]), seqs[g], lens[g]);
          e = hipMemcpyAsync(d_seq + offs[g0], pin, span, hipMemcpyHostToDevice, up_stream);
          if (e == hipSuccess) e = hipEventRecord(c->pack_ev[b], up_stream);
          c->pack_used[b] = true;
        }
      } else {
        for (size_t g = g0; g < g1 && e == hipSuccess; ++g)
          if (lens[g]) e = hipMemcpyAsync(d_seq + offs[g], seqs[g], lens[g], hipMemcpyHostToDevice, up_stream);
      }
      if (e == hipSuccess && !one_stream) e = hipEventRecord(c->copy_events[k], up_stream);
      std::lock_guard<std::mutex> lk(mu);
      if (e != hipSuccess) copy_err = e;
      queued = e == hipSuccess ? k + 1 : n_chunks;  // on error release the consumer, which then reports it
      cv.notify_all();
      if (e != hipSuccess) return;
    }
  };
  std::thread uploader;
  if (n_chunks > 1) uploader = std::thread(upload, (size_t)0);
  else upload(0);
  s = HG_OK;
  for (size_t k = 0; k < n_chunks && s == HG_OK; ++k) {
    {
      std::unique_lock<std::mutex> lk(mu);
      cv.wait(lk, [&] { return queued > k; });
      if (copy_err != hipSuccess) {
        s = hg_fail(c, HG_ERR_HIP, std::string("sequence upload: ") + hipGetErrorString(copy_err));
        break;
      }
    }
    const size_t g0 = cut[k], m = cut[k + 1] - g0;
    hipError_t e = one_stream ? hipSuccess : hipStreamWaitEvent(c->stream, c->copy_events[k], 0);
    if (e != hipSuccess) {
      s = hg_fail(c, HG_ERR_HIP, std::string("hipStreamWaitEvent: ") + hipGetErrorString(e));
      break;
    }
    if (sub_packed[k])
      s = hg_sketch_batch_dev_packed(c, d_seq, boffs.data() + g0, l64.data() + g0, m, p, d_hv + g0 * (size_t)p->hv_d, d_n2 + g0,
                                     d_nh + g0);
    else
      s = hg_sketch_batch_dev(c, d_seq, offs.data() + g0, l64.data() + g0, m, p, d_hv + g0 * (size_t)p->hv_d, d_n2 + g0,
                              d_nh + g0);
    if (s != HG_OK) break;
    e = hipMemcpyAsync(hv_out + g0 * (size_t)p->hv_d, d_hv + g0 * (size_t)p->hv_d, m * (size_t)p->hv_d * sizeof(int16_t),
                       hipMemcpyDeviceToHost, c->stream);
    if (e != hipSuccess) s = hg_fail(c, HG_ERR_HIP, std::string("hipMemcpyAsync: ") + hipGetErrorString(e));
  }
  if (uploader.joinable()) uploader.join();
  if (!one_stream) (void)hipStreamSynchronize(c->copy_stream);
  if (s != HG_OK) {
    (void)hipStreamSynchronize(c->stream);
    return s;
  }
  HG_HIP(c, hipMemcpyAsync(norm2_out, d_n2, n * sizeof(int32_t), hipMemcpyDeviceToHost, c->stream));
  HG_HIP(c, hipMemcpyAsync(nhash_out, d_nh, n * sizeof(uint32_t), hipMemcpyDeviceToHost, c->stream));
  HG_HIP(c, hipStreamSynchronize(c->stream));
  return HG_OK;
}

extern "C" hg_status hg_kmer_hash_sample(hg_ctx *c, const uint8_t *seq, size_t n_bps, uint32_t ksize,
                                         uint64_t threshold, uint64_t seed, int canonical, uint32_t norm_mode,
                                         uint64_t *out_hashes, size_t cap, size_t *n_out) {
  if (!c) return HG_ERR_INVALID;
  if (!n_out) return hg_fail(c, HG_ERR_INVALID, "n_out == NULL");
  *n_out = 0;
  if (ksize < 1) return hg_fail(c, HG_ERR_INVALID, "ksize must be >= 1");
  if (ksize > 255) return hg_fail(c, HG_ERR_UNSUPPORTED, "ksize must be <= 255 (the reference's -k is u8)");
  if (norm_mode > HG_NORM_U2T) return hg_fail(c, HG_ERR_INVALID, "bad norm mode");
  if (n_bps && !seq) return hg_fail(c, HG_ERR_INVALID, "NULL sequence");
  if (n_bps < ksize) return HG_OK;
  HG_HIP(c, hipSetDevice(c->device));
  const std::vector<uint64_t> offs{0}, l64{n_bps};
  hg_status s = hg_ensure(c, c->w_seq, n_bps + 64);
  if (s != HG_OK) return s;
  // over the link as ASCII, or 2-bit packed by this thread when the link is shared with other calls (pack_single())
  HostfedCall in_flight;
  const SingleChoice single = pack_single(c, seq, n_bps, in_flight.others);
  const bool packed = single.packed;
  if (packed) {
    HG_HIP(c, pack_buf_for(c, 0, hg_pack2_size(n_bps)));
    const auto tp0 = std::chrono::steady_clock::now();
    hg_pack2_piece(seq, n_bps, norm_mode, static_cast<uint8_t *>(c->pack_buf[0]), 0, n_bps);  // (the whole genome as one piece)
    pack_measured(single, n_bps, std::chrono::duration<double>(std::chrono::steady_clock::now() - tp0).count());
    HG_HIP(c, hipMemcpyAsync(c->w_seq.p, c->pack_buf[0], hg_pack2_size(n_bps), hipMemcpyHostToDevice, c->stream));
    HG_HIP(c, hipEventRecord(c->pack_ev[0], c->stream));
    c->pack_used[0] = true;
  } else {
    HG_HIP(c, hipMemcpyAsync(c->w_seq.p, seq, n_bps, hipMemcpyHostToDevice, c->stream));
  }
  // capacity heuristic wants "scaled"; derive it from the threshold (threshold = MAX / scaled)
  uint64_t scaled = threshold ? UINT64_MAX / threshold : UINT64_MAX;
  if (scaled < 1) scaled = 1;
  BatchPlan pl;
  uint32_t *d_nd = nullptr;
  SampleFetch fetch;
  fetch.max_hashes = out_hashes ? cap : 0;
  s = sample_batch(c, static_cast<uint8_t *>(c->w_seq.p), offs.data(), l64.data(), 1, ksize, threshold, scaled,
                   seed, canonical != 0, norm_mode, pl, &d_nd, packed, nullptr, &fetch);
  if (s != HG_OK) return s;
  if (fetch.valid) {  // count and hashes came back with sample_batch's own synchronisation
    *n_out = fetch.nd;
    if (fetch.nd) std::memcpy(out_hashes, fetch.h_hashes, fetch.nd * sizeof(uint64_t));
    return HG_OK;
  }
  uint32_t nd = 0;
  HG_HIP(c, hipMemcpyAsync(&nd, d_nd, sizeof nd, hipMemcpyDeviceToHost, c->stream));
  HG_HIP(c, hipStreamSynchronize(c->stream));
  *n_out = nd;
  if (nd > cap) return hg_fail(c, HG_ERR_CAPACITY, "out_hashes too small");
  if (nd) {
    if (!out_hashes) return hg_fail(c, HG_ERR_INVALID, "out_hashes == NULL");
    HG_HIP(c, hipMemcpyAsync(out_hashes, static_cast<uint64_t *>(c->w_hits.p) + pl.meta[0].hit_off,
                             nd * sizeof(uint64_t), hipMemcpyDeviceToHost, c->stream));
    HG_HIP(c, hipStreamSynchronize(c->stream));
  }
  return HG_OK;
}

extern "C" hg_status hg_hv_encode(hg_ctx *c, const uint64_t *hashes, size_t n, uint32_t hv_d, uint32_t hv_layout,
                                  int16_t *hv_out, int32_t *norm2_out) {
  if (!c) return HG_ERR_INVALID;
  if (hv_d == 0 || hv_d > 32768) return hg_fail(c, HG_ERR_UNSUPPORTED, "hv_d must be in 1..32768");
  if (hv_layout > HG_LAYOUT_AVX2) return hg_fail(c, HG_ERR_INVALID, "bad layout");
  if ((n && !hashes) || !hv_out || !norm2_out) return hg_fail(c, HG_ERR_INVALID, "NULL argument");
  if (n > 0xFFFFFFF0ull) return hg_fail(c, HG_ERR_UNSUPPORTED, "too many hashes");
  HG_HIP(c, hipSetDevice(c->device));
  hg_status s;
  if ((s = hg_ensure(c, c->w_gmeta, sizeof(hg_genome_meta))) != HG_OK) return s;
  if ((s = hg_ensure(c, c->w_hits, (n + 1) * sizeof(uint64_t))) != HG_OK) return s;
  if ((s = hg_ensure(c, c->w_cnt, 2 * sizeof(uint32_t))) != HG_OK) return s;
  if ((s = hg_ensure(c, c->w_hv, (size_t)hv_d * sizeof(int16_t) + 64)) != HG_OK) return s;
  c->plan_valid = false;  // w_gmeta is about to be overwritten
  hg_genome_meta m{};
  m.hit_off = 0, m.hit_cap = (uint32_t)n;
  const uint32_t nd = (uint32_t)n;
  auto *d_hv = static_cast<int16_t *>(c->w_hv.p);
  auto *d_n2 = reinterpret_cast<int32_t *>(static_cast<uint8_t *>(c->w_hv.p) + (((size_t)hv_d * 2 + 15) & ~(size_t)15));
  HG_HIP(c, hipMemcpyAsync(c->w_gmeta.p, &m, sizeof m, hipMemcpyHostToDevice, c->stream));
  HG_HIP(c, hipMemcpyAsync(c->w_cnt.p, &nd, sizeof nd, hipMemcpyHostToDevice, c->stream));
  if (n) HG_HIP(c, hipMemcpyAsync(c->w_hits.p, hashes, n * sizeof(uint64_t), hipMemcpyHostToDevice, c->stream));
  HG_HIP(c, hg_launch_encode(c->stream, static_cast<hg_genome_meta *>(c->w_gmeta.p), 1,
                             static_cast<uint64_t *>(c->w_hits.p), static_cast<uint32_t *>(c->w_cnt.p), hv_d,
                             hv_layout, d_hv, d_n2, nullptr, nd));
  HG_HIP(c, hipMemcpyAsync(hv_out, d_hv, (size_t)hv_d * sizeof(int16_t), hipMemcpyDeviceToHost, c->stream));
  HG_HIP(c, hipMemcpyAsync(norm2_out, d_n2, sizeof(int32_t), hipMemcpyDeviceToHost, c->stream));
  HG_HIP(c, hipStreamSynchronize(c->stream));
  return HG_OK;
}
