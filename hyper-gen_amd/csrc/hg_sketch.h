// hg_sketch.h -- seams between the translation units of the sketch path (not installed):
//   hg_sketch_plan.hip  batch geometry -> hit regions + work items; the plan kept by the ctx; its upload
//   hg_sketch_step.hip  the sync-free step (hash + sample -> sort / unique -> encode queued back to back, one check word
//                       read a call late) and hg_sketch_resolve
//   hg_sketch_rare.hip  the synchronous path: counters read back, overflow retry, multi-workgroup sorts, split encode
//   hg_api_sketch.hip   the C ABI entry points (device-resident, host-fed, one genome per call)
#pragma once
#include <utility>

#include "hg_internal.h"

// Host tables of one batch: what the kernels read (uploaded to w_gmeta / w_items) and what the synchronous path learns
// from the raw counters.
struct hg_batch_tables {
  std::vector<hg_genome_meta> meta;   // (a plan taken from the ctx's cache fills hit_cap / hit_off only)
  std::vector<uint32_t> item_genome;  // work item -> genome (empty for a cached plan)
  std::vector<uint32_t> group_first;  // workgroup -> its first work item, n_groups + 1 entries (empty for a cached plan / k > 32)
  size_t n_groups = 0;
  uint64_t total_slots = 0;
  uint32_t max_cap = 0, max_expect = 0;
  size_t n_items = 0;
  uint32_t max_hits = ~0u;  // largest stored raw hit count of the batch (upper bound of the distinct counts)
  std::vector<std::pair<uint32_t, uint32_t>> big;  // (genome, stored raw hits) with more than HG_ENC_SLAB hits
};

hg_status hg_check_sketch_params(hg_ctx *c, const hg_sketch_params *p);
// where the group table lies in w_items: behind the n_items item words, 16-byte aligned
inline size_t hg_plan_group_offset(size_t n_items) { return (n_items + 3) & ~(size_t)3; }
inline const uint32_t *hg_plan_group_table(const hg_ctx *c, size_t n_items, size_t n_groups) {
  return n_groups ? static_cast<const uint32_t *>(c->w_items.p) + hg_plan_group_offset(n_items) : nullptr;
}

// want_caps: optional per-genome minimum capacities (retry after overflow)
hg_status hg_plan_build(hg_ctx *c, const uint64_t *offsets, const uint64_t *lens, size_t n, uint32_t ksize, uint64_t scaled,
                        const std::vector<uint32_t> *want_caps, hg_batch_tables &t, const uint64_t *mask_offs);
// the ctx's cached plan has this geometry (its tables are on the device)
bool hg_plan_matches(const hg_ctx *c, const uint64_t *offsets, const uint64_t *lens, const uint64_t *mask_offs, size_t n,
                     uint32_t ksize, uint64_t scaled, bool packed);
// with_meta = false: the totals only (the sync-free step reads no per-genome record on the host: 400 000 genomes are 16 MB of them)
void hg_plan_tables_from_cache(const hg_sketch_plan &pl, size_t n, hg_batch_tables &t, bool with_meta = true);
// Sends a freshly built plan's tables to w_gmeta / w_items through the page-locked plan staging (stream-ordered,
// returns at once) and makes it the ctx's cached plan.
hg_status hg_plan_upload(hg_ctx *c, const hg_batch_tables &t, const uint64_t *offsets, const uint64_t *lens,
                         const uint64_t *mask_offs, size_t n, uint32_t ksize, uint64_t scaled, bool packed);

// ASCII genomes -> hg_pack2 blobs on the device (synchronises the stream before and after: it uses the ctx's scratch)
hg_status hg_pack_batch(hg_ctx *c, const uint8_t *d_seq, const uint64_t *seq_offs, const uint64_t *lens, size_t n,
                        uint32_t norm_mode, uint8_t *d_blobs, const uint64_t *blob_offs);

// One-genome callers that want the sorted hash list on the host (hg_kmer_hash_sample): the distinct count and the first
// max_hashes hashes ride back with the counter copy the synchronous path waits for anyway -- one synchronisation per call
// instead of three.  valid is set when the list the LDS sort produced is final (no overflow, no second sort pass).
struct hg_sample_fetch {
  size_t max_hashes = 0;
  const uint64_t *h_hashes = nullptr;  // in the ctx's page-locked scratch: consume before the next call on the ctx
  uint32_t nd = 0;
  bool valid = false;
};

// The synchronous path.  Runs hash + sample and sort / unique; on return (stream synchronised) the device hit buffer holds
// each genome's ascending distinct hashes at t.meta[g].hit_off and *d_ndistinct_out the counts.
hg_status hg_sample_batch_sync(hg_ctx *c, const uint8_t *d_seq, const uint64_t *offsets, const uint64_t *lens, size_t n,
                               uint32_t ksize, uint64_t threshold, uint64_t scaled_for_cap, uint64_t seed, bool canonical,
                               uint32_t norm_mode, hg_batch_tables &t, uint32_t **d_ndistinct_out, bool packed,
                               const uint64_t *mask_offs, hg_sample_fetch *fetch);
// ... followed by the encoders (several workgroups for the genomes with very large sets); returns with the encoders
// queued, everything before them finished.
hg_status hg_sketch_batch_sync(hg_ctx *c, const uint8_t *d_seq, const uint64_t *offsets, const uint64_t *lens, size_t n,
                               const hg_sketch_params *p, int16_t *d_hv, int32_t *d_norm2, uint32_t *d_nhash, bool packed,
                               const uint64_t *mask_offs);

// One sketch step on device-resident genomes: sync-free when the batch allows it, else the synchronous path.
hg_status hg_sketch_step(hg_ctx *c, const uint8_t *d_seq, const uint64_t *offsets, const uint64_t *lens, size_t n,
                         const hg_sketch_params *p, int16_t *d_hv, int32_t *d_norm2, uint32_t *d_nhash, bool packed,
                         const uint64_t *mask_offs);
