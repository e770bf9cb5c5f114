// hg_sketch_rare.hip -- the synchronous sketch path: the per-genome raw counters are read back between sort and encode, and
// the host takes what the sync-free step (hg_sketch_step.hip) cannot do on the device: growing hit regions after an
// overflow, the multi-workgroup sorts of hash sets beyond the one-workgroup sort, the split encode of very large sets.
// Taken for batches whose genomes are EXPECTED to need it (>= 10 Mbp at scaled = 1 500), for the re-run of a step whose
// check word asked for it, and by hg_kmer_hash_sample (its result goes to the host anyway).
#include <algorithm>
#include <cstring>

#include "hg_sketch.h"

// Sorts + de-duplicates the genomes whose sampled hash count exceeds what one workgroup sorts in LDS.
// h_cnt: raw per-genome counters (host copy).  Synchronises the stream when it had work to do.
static hg_status sort_large_sets(hg_ctx *c, const hg_batch_tables &pl, const uint32_t *h_cnt, size_t n, uint64_t threshold,
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

hg_status hg_sample_batch_sync(hg_ctx *c, const uint8_t *d_seq, const uint64_t *offsets, const uint64_t *lens, size_t n,
                               uint32_t ksize, uint64_t threshold, uint64_t scaled_for_cap, uint64_t seed, bool canonical,
                               uint32_t norm_mode, hg_batch_tables &pl, uint32_t **d_ndistinct_out, bool packed,
                               const uint64_t *mask_offs, hg_sample_fetch *fetch) {
  if (n > 0x7FFFFFFFull) return hg_fail(c, HG_ERR_UNSUPPORTED, "more than 2^31 genomes in one batch");
  ++c->n_sync_steps;
  std::vector<uint32_t> want;
  for (int attempt = 0; attempt < 3; ++attempt) {
    hg_status s;
    // same geometry as the previous call (typical for a stream of equally shaped batches): the
    // work-item table and the per-genome records are still on the device
    const bool reuse = attempt == 0 && hg_plan_matches(c, offsets, lens, mask_offs, n, ksize, scaled_for_cap, packed);
    if (reuse) {
      hg_plan_tables_from_cache(*c->plan, n, pl);
    } else {
      if ((s = hg_plan_build(c, offsets, lens, n, ksize, scaled_for_cap, attempt ? &want : nullptr, pl, mask_offs)) != HG_OK) return s;
      if ((s = hg_plan_upload(c, pl, offsets, lens, mask_offs, n, ksize, scaled_for_cap, packed)) != HG_OK) return s;
    }
    const size_t n_items = pl.n_items;
    if ((s = hg_ensure(c, c->w_hits, pl.total_slots * sizeof(uint64_t) + 16)) != HG_OK) return s;
    if ((s = hg_ensure(c, c->w_cnt, (2 * n + 16) * sizeof(uint32_t) + 16)) != HG_OK) return s;
    // page-locked scratch of this path: the counters, the fetch block (count + first hashes of a one-genome call), the redo list
    const size_t cnt_bytes = (n * sizeof(uint32_t) + 63) & ~(size_t)63;
    const size_t fetch_n = (fetch && n == 1) ? std::min<size_t>({fetch->max_hashes, pl.meta[0].hit_cap, (size_t)1 << 16}) : 0;
    const size_t fetch_bytes = (fetch && n == 1) ? ((64 + fetch_n * 8 + 63) & ~(size_t)63) : 0;
    const size_t fetch_off = cnt_bytes, redo_off = cnt_bytes + fetch_bytes;
    if ((s = hg_ensure_pinned(c, redo_off + cnt_bytes + 64)) != HG_OK) return s;
    auto *d_meta = static_cast<hg_genome_meta *>(c->w_gmeta.p);
    auto *d_items = static_cast<uint32_t *>(c->w_items.p);
    auto *d_hits = static_cast<uint64_t *>(c->w_hits.p);
    auto *d_cnt = static_cast<uint32_t *>(c->w_cnt.p);
    uint32_t *d_nd = d_cnt + n;
    auto *h_cnt = static_cast<uint32_t *>(c->h_pin);
    HG_HIP(c, hipMemsetAsync(d_cnt, 0, 2 * n * sizeof(uint32_t), c->stream));
    {
      hg_timed tm(c, HG_T_KMER);
      c->last_kernel[HG_T_KMER] = hg_kmer_kernel_name(ksize, canonical, packed);
      HG_HIP(c, hg_launch_kmer_sample(c->stream, d_seq, d_meta, d_items, (uint32_t)n_items, ksize, threshold,
                                      seed, canonical, norm_mode, d_hits, d_cnt, packed,
                                      hg_plan_group_table(c, n_items, pl.n_groups), (uint32_t)pl.n_groups));
    }
    uint32_t sort_cap = pl.max_cap;
    {
      // The LDS sort is sized by the genomes' CAPACITIES (twice the expected count + 1 024: 64 KiB for a 5 Mbp genome,
      // two workgroups per CU).  When the plan is a repeat, the counts of its last run are known: size by those (+ 12.5 %,
      // 32 KiB -> five workgroups per CU); a genome that outgrows it is left to the large-set path below, as always.
      const uint32_t seen = reuse ? c->plan->max_hits : 0;
      if (seen) sort_cap = (uint32_t)std::min<uint64_t>(sort_cap, (uint64_t)seen + seen / 8 + 16);
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
    c->plan_upload_pending = false;  // (whatever was uploaded has passed)
    bool overflow = false;
    want.assign(n, 0);
    for (size_t g = 0; g < n; ++g)
      if (h_cnt[g] > pl.meta[g].hit_cap) overflow = true, want[g] = h_cnt[g];
    if (!overflow) {
      if (hg_sort_lds_keys(sort_cap) < hg_sort_lds_keys(pl.max_cap)) {
        // the count-sized sort left out every genome that grew past its size: those again, with the capacity-sized one
        // (the list goes up through the page-locked scratch on the ctx's own stream, like every other command of this path)
        uint32_t *h_redo = reinterpret_cast<uint32_t *>(static_cast<uint8_t *>(c->h_pin) + redo_off);
        size_t n_redo = 0;
        const uint32_t keys = hg_sort_lds_keys(sort_cap);
        for (size_t g = 0; g < n; ++g) {
          const uint32_t cnt = std::min(h_cnt[g], pl.meta[g].hit_cap);
          if (cnt > keys && cnt <= HG_SORT_LDS_MAX_KEYS) h_redo[n_redo++] = (uint32_t)g;
        }
        if (n_redo) {
          if ((s = hg_ensure(c, c->w_redo, n_redo * 4 + 64)) != HG_OK) return s;
          HG_HIP(c, hipMemcpyAsync(c->w_redo.p, h_redo, n_redo * 4, hipMemcpyHostToDevice, c->stream));
          hg_timed tm(c, HG_T_SORT);
          HG_HIP(c, hg_launch_sort_unique_todo(c->stream, d_meta, static_cast<uint32_t *>(c->w_redo.p), (uint32_t)n_redo,
                                               d_hits, d_cnt, d_nd, pl.max_cap, threshold));
          HG_HIP(c, hipStreamSynchronize(c->stream));  // (rare path; the next call may rewrite the scratch at once)
        }
      }
      // hash sets beyond the LDS sort: bucketed multi-workgroup sort (or, where that cannot work, in place)
      if ((s = sort_large_sets(c, pl, h_cnt, n, threshold, d_hits, d_cnt, d_nd)) != HG_OK) return s;
      pl.big.clear();
      pl.max_hits = 0;
      for (size_t g = 0; g < n; ++g) {
        const uint32_t cnt = std::min(h_cnt[g], pl.meta[g].hit_cap);
        pl.max_hits = std::max(pl.max_hits, cnt);
        if (cnt > HG_ENC_SLAB) pl.big.emplace_back((uint32_t)g, cnt);
      }
      if (c->plan) c->plan->max_hits = pl.max_hits;
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
    c->plan.reset();
  }
  return hg_fail(c, HG_ERR_HIP, "hit buffer overflow persisted after resizing");
}

hg_status hg_sketch_batch_sync(hg_ctx *c, const uint8_t *d_seq, const uint64_t *offsets, const uint64_t *lens, size_t n,
                               const hg_sketch_params *p, int16_t *d_hv, int32_t *d_norm2, uint32_t *d_nhash, bool packed,
                               const uint64_t *mask_offs) {
  hg_batch_tables pl;
  uint32_t *d_nd = nullptr;
  const uint64_t threshold = UINT64_MAX / p->scaled;  // src/sketch.rs:73
  hg_status s = hg_sample_batch_sync(c, d_seq, offsets, lens, n, p->ksize, threshold, p->scaled, p->seed, p->canonical != 0,
                                     p->norm_mode, pl, &d_nd, packed, mask_offs, nullptr);
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
