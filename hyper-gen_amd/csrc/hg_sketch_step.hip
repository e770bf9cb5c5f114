// hg_sketch_step.hip -- one sketch step on device-resident genomes with the host out of the loop.
//
// The step is what src/sketch.rs:35-56 does per file -- hash + sample (extract_kmer_hash), the set (HashSet<u64>), the
// encode and its norm -- for all genomes of a batch: k-mer kernel -> sort / unique -> encoders.  The host has three
// questions between those kernels: did a genome outgrow its hit region (then the regions must grow and the batch run
// again), does a hash set exceed the one-workgroup sort (then the multi-workgroup sorts take it), how large is the largest
// set (which encoder).  Reading the counters back to answer them cost a blocking round trip in the middle of every step
// and three n-long host loops behind it: 0.04 ms per 1 000-genome step on a quiet host, 0.4-2 ms on a busy one.
//
// Here the device answers them.  The sort kernel itself sets a bit in the step's flag word for a genome it cannot take and
// marks that genome's count HG_NHASH_PENDING (the encoders skip it); the genomes a count-sized sort left out are found in
// the counters by the second sort launch's own workgroups; the encoders are launched for the largest set the plan allows.
// The flag word travels to a page-locked slot behind the last kernel, and the host reads it when the NEXT call on the ctx
// (or hg_ctx_sync) arrives -- after that call's own kernels are queued, so the device never waits for the host.  A step
// whose word is not zero is run again through the synchronous path (hg_sketch_rare.hip); that is rare by construction
// (a sampled k-mer repeated thousands of times, or a set 2.4x its expected size).
#include <algorithm>
#include <atomic>
#include <chrono>
#include <cstring>
#include <thread>

#include <unistd.h>

#include "hg_sketch.h"

namespace {

// the sort launched first is sized for this many raw hits per genome: what the plan's last synchronous run saw, or the
// expected count, + 12.5 % (seven standard deviations at 3 333)
uint32_t first_sort_cap(const hg_sketch_plan &pl) {
  const uint64_t base = pl.max_hits ? pl.max_hits : pl.max_expect;
  // (a batch of genomes that sample at most ~35 k-mers each -- up to 50 kbp at scaled = 1 500 -- stays within the 64 keys one
  // WAVE sorts: sort_unique_wave_kernel, four genomes per workgroup)
  const uint64_t margin = base + base / 8 + 24 <= 64 ? 24 : 64;
  return (uint32_t)std::min<uint64_t>(pl.max_cap, base + base / 8 + margin);
}

// the batch can go without the host: every genome is EXPECTED to fit the one-workgroup sort (those that do not after all
// are reported by the flag word)
bool step_can_be_sync_free(const hg_ctx *c, const hg_batch_tables &t, size_t n) {
  if (c->dbg_sketch_path == "sync" || c->dbg_sort_buckets) return false;
  if (n == 0 || n > 0x7FFFFFFFull) return false;
  return (uint64_t)t.max_expect + t.max_expect / 8 + 64 <= HG_SORT_LDS_MAX_KEYS;
}

// Waits for the check word of a queued step.  The device is busy with work queued BEHIND that step when this is called
// from the next step (the common case), so nothing is gained by a hot spin: a short one, then naps.
hg_status wait_check_word(hg_ctx *c, int slot, uint32_t seq, uint32_t *flags) {
  volatile uint32_t *w = c->h_chk + 16 * slot;
  const auto t0 = std::chrono::steady_clock::now();
  for (uint32_t spins = 0;; ++spins) {
    if (w[1] == seq) break;
    if (spins < 4096) {
#if defined(__x86_64__) || defined(__i386__)
      __builtin_ia32_pause();
#else
      std::this_thread::yield();
#endif
      continue;
    }
    ::usleep(50);
    if ((spins & 0x3ff) == 0 && std::chrono::steady_clock::now() - t0 > std::chrono::seconds(2)) {
      // a step does not take this long: let the runtime say whether the stream is alive, then look once more
      const hipError_t e = hipStreamSynchronize(c->stream);
      if (e != hipSuccess) return hg_fail(c, HG_ERR_HIP, std::string("hipStreamSynchronize: ") + hipGetErrorString(e));
      if (w[1] != seq) return hg_fail(c, HG_ERR_HIP, "sketch step: the stream finished without publishing its check word");
      break;
    }
  }
  std::atomic_thread_fence(std::memory_order_acquire);
  *flags = w[0];
  return HG_OK;
}

hg_status redo_step(hg_ctx *c, const hg_sketch_pending &pd) {
  ++c->n_redone_steps;
  const hg_sketch_plan &pl = *pd.plan;
  return hg_sketch_batch_sync(c, pd.d_seq, pl.offs.data(), pl.lens.data(), pl.offs.size(), &pd.p, pd.d_hv, pd.d_norm2, pd.d_nhash,
                              pl.packed, pl.masks.empty() ? nullptr : pl.masks.data());
}

}  // namespace

hg_status hg_sketch_resolve(hg_ctx *c, bool *redone) {
  if (redone) *redone = false;
  if (!c->pending.active) return HG_OK;
  const hg_sketch_pending pd = std::move(c->pending);
  c->pending = hg_sketch_pending{};
  uint32_t flags = 0;
  hg_status s = wait_check_word(c, pd.slot, pd.seq, &flags);
  if (s != HG_OK || !flags) return s;
  if (redone) *redone = true;
  return redo_step(c, pd);
}

hg_status hg_sketch_step(hg_ctx *c, const uint8_t *d_seq, const uint64_t *offsets, const uint64_t *lens, size_t n,
                         const hg_sketch_params *p, int16_t *d_hv, int32_t *d_norm2, uint32_t *d_nhash, bool packed,
                         const uint64_t *mask_offs) {
  // (entered with the previous step's check word unread: it is looked at below, behind this step's launches)
  hg_status s;
  hg_batch_tables t;
  const bool reuse = hg_plan_matches(c, offsets, lens, mask_offs, n, p->ksize, p->scaled, packed);
  if (reuse) hg_plan_tables_from_cache(*c->plan, n, t, false);
  else if ((s = hg_plan_build(c, offsets, lens, n, p->ksize, p->scaled, nullptr, t, mask_offs)) != HG_OK) return s;
  if (!step_can_be_sync_free(c, t, n)) {
    if ((s = hg_sketch_resolve(c)) != HG_OK) return s;
    return hg_sketch_batch_sync(c, d_seq, offsets, lens, n, p, d_hv, d_norm2, d_nhash, packed, mask_offs);
  }
  if (!c->h_chk) {
    void *q = nullptr;
    // (coherent = fine-grained: the device's writes and the fence between flags and sequence number reach the polling host)
    const hipError_t e = hipHostMalloc(&q, 32 * sizeof(uint32_t), hipHostMallocCoherent);
    if (e != hipSuccess) return hg_fail(c, HG_ERR_OOM, std::string("hipHostMalloc: ") + hipGetErrorString(e));
    c->h_chk = static_cast<uint32_t *>(q);
    std::memset(c->h_chk, 0, 32 * sizeof(uint32_t));
  }
  if (!reuse && (s = hg_plan_upload(c, t, offsets, lens, mask_offs, n, p->ksize, p->scaled, packed)) != HG_OK) return s;
  if ((s = hg_ensure(c, c->w_hits, t.total_slots * sizeof(uint64_t) + 16)) != HG_OK) return s;
  if ((s = hg_ensure(c, c->w_cnt, (2 * n + 16) * sizeof(uint32_t) + 16)) != HG_OK) return s;
  auto *d_meta = static_cast<hg_genome_meta *>(c->w_gmeta.p);
  auto *d_items = static_cast<uint32_t *>(c->w_items.p);
  auto *d_hits = static_cast<uint64_t *>(c->w_hits.p);
  auto *d_cnt = static_cast<uint32_t *>(c->w_cnt.p);
  uint32_t *d_nd = d_cnt + n, *d_flags = d_cnt + 2 * n;
  const uint64_t threshold = UINT64_MAX / p->scaled;  // src/sketch.rs:73
  const uint32_t seq = ++c->chk_seq ? c->chk_seq : ++c->chk_seq;  // never 0
  const int slot = (int)(seq & 1u);

  HG_HIP(c, hipMemsetAsync(d_cnt, 0, (2 * n + 16) * sizeof(uint32_t), c->stream));
  {
    hg_timed tm(c, HG_T_KMER);
    c->last_kernel[HG_T_KMER] = hg_kmer_kernel_name(p->ksize, p->canonical != 0, packed);
    HG_HIP(c, hg_launch_kmer_sample(c->stream, d_seq, d_meta, d_items, (uint32_t)t.n_items, p->ksize, threshold, p->seed,
                                    p->canonical != 0, p->norm_mode, d_hits, d_cnt, packed,
                                    hg_plan_group_table(c, t.n_items, t.n_groups), (uint32_t)t.n_groups));
  }
  {
    // The LDS sort sized by the genomes' CAPACITIES (twice the expected count + 1 024) takes 64 KiB for a 5 Mbp genome: two
    // workgroups per CU.  Sized by the counts to be expected it takes 32 KiB -- five per CU --, and the few genomes that
    // outgrow it are picked up by the second launch (capacity-sized; its workgroups leave at once when there are none).
    const uint32_t sort_cap = first_sort_cap(*c->plan);
    hg_timed tm(c, HG_T_SORT, HG_T_KMER);
    HG_HIP(c, hg_launch_sort_unique(c->stream, d_meta, (uint32_t)n, d_hits, d_cnt, d_nd, sort_cap, threshold, d_flags));
    HG_HIP(c, hg_launch_sort_unique_rest(c->stream, d_meta, (uint32_t)n, d_hits, d_cnt, d_nd, sort_cap, t.max_cap, threshold));
  }
  {
    hg_timed tm(c, HG_T_ENCODE, HG_T_SORT);
    HG_HIP(c, hg_launch_encode(c->stream, d_meta, (uint32_t)n, d_hits, d_nd, p->hv_d, p->hv_layout, d_hv, d_norm2, nullptr,
                               std::min<uint32_t>(t.max_cap, HG_SORT_LDS_MAX_KEYS)));
    HG_HIP(c, hg_launch_sketch_finish(c->stream, d_nd, d_nhash, (uint32_t)n, d_flags, c->h_chk + 16 * slot, seq));
  }
  ++c->n_fast_steps;

  // this step is queued: now the previous one's check word (its slot is the other one)
  hg_sketch_pending cur;
  cur.active = true, cur.seq = seq, cur.slot = slot, cur.plan = c->plan;
  cur.d_seq = d_seq, cur.p = *p, cur.d_hv = d_hv, cur.d_norm2 = d_norm2, cur.d_nhash = d_nhash;
  bool redone = false;
  if ((s = hg_sketch_resolve(c, &redone)) != HG_OK) return s;
  if (redone) {
    // the previous step was run again BEHIND this one: if the two share output buffers, the older results now lie on top.
    // Run this one again too, so that the buffers hold what the order of the calls says.
    return redo_step(c, cur);
  }
  c->pending = std::move(cur);
  return HG_OK;
}
