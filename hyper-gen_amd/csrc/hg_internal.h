// hg_internal.h -- shared declarations of the library's translation units (not installed).
#pragma once
#include <hip/hip_runtime.h>
#include <stddef.h>
#include <stdint.h>

#include <memory>
#include <string>
#include <vector>

#include "../../include/hypergen.h"

// ---- sketch batch plans (hg_sketch_plan.hip) ----------------------------------------------------------------
// Geometry of one sketch batch as the k-mer kernel sees it: hit regions and work items per genome.  Kept by the ctx
// between calls (the device tables stay valid for a batch of the same shape) and by a queued step until its check
// word has been read (the redo of a step needs the offsets it was queued with).
struct hg_sketch_plan {
  std::vector<uint64_t> offs, lens, masks;  // the caller's arrays (masks empty: bitmaps directly behind the codes)
  std::vector<uint32_t> caps;               // hit_cap per genome
  uint32_t ksize = 0;
  bool packed = false;
  uint64_t scaled = 0, total_slots = 0;
  uint32_t max_cap = 0;
  uint32_t max_expect = 0;  // largest expected sampled count (k-mer starts / scaled) of a genome
  uint32_t max_hits = 0;    // largest raw hit count the plan's last synchronous run saw (0: unknown)
  size_t n_items = 0;
  size_t n_groups = 0;      // workgroups of the k-mer launch: groups of consecutive work items (w_items holds the table behind the items)
};
struct hg_sketch_pending {
  bool active = false;
  uint32_t seq = 0;
  int slot = 0;
  std::shared_ptr<const hg_sketch_plan> plan;
  const uint8_t *d_seq = nullptr;
  hg_sketch_params p{};
  int16_t *d_hv = nullptr;
  int32_t *d_norm2 = nullptr;
  uint32_t *d_nhash = nullptr;
};

// ---- error plumbing ----------------------------------------------------------------
struct hg_ctx {
  int device = 0;
  hipStream_t own_stream = nullptr;
  hipStream_t stream = nullptr;  // own_stream or a borrowed one
  std::string err;
  int n_cu = 256;

  // growable device workspaces (never shrunk; freed in hg_ctx_destroy)
  struct Buf {
    void *p = nullptr;
    size_t cap = 0;
  };
  Buf w_items;    // uint32 work-item -> genome table
  Buf w_gmeta;    // per-genome metadata (hg_genome_meta)
  Buf w_hits;     // sampled hashes, per-genome regions
  Buf w_cnt;      // per-genome raw hit counters + distinct counts
  Buf w_hits2;    // scratch copy of the hit buffer (bucketed sort of large hash sets)
  Buf w_lsort;    // job / chunk / bucket tables of the bucketed sort
  Buf w_redo;     // genomes the count-sized LDS sort skipped (see sample_batch)
  Buf w_pk;       // hg_pack2 blobs of a batch that arrived as ASCII (debug hook "kmer_input" = "packed", hg_pack2_batch_dev scratch)
  Buf w_pktab;    // ... and the per-genome offset tables of the packing kernel
  Buf w_seq;      // staged sequences (host entry points)
  Buf w_hv;       // staged HV output (host entry points)
  Buf w_misc;     // small scalars (hit counters of dist, flags)
  const void *misc_zeroed = nullptr;  // == w_misc.p while its 64 bytes are zero (or being zeroed on `stream`): a call leaves them so for the next
  Buf w_f16a;     // f16 copies of the HV matrices for the MFMA path
  Buf w_f16b;
  Buf w_stats;    // per-row |max| / block sums of squares
  Buf w_ani;      // staged ANI output (host entry points)
  Buf w_hv2;      // staged second HV matrix (host dist)
  Buf w_n2a, w_n2b;
  Buf w_i8a, w_i8b, w_i8misc;  // i8 operand copies (+ extra columns), row info / outlier list / column maps
  struct I8Pad {               // which zero padding rows w_i8a / w_i8b hold since the last dist call (hg_run_dist)
    const void *ptr = nullptr;
    uint32_t rows = 0, padded = 0, pitch = 0;
  } i8_pad[2];
  Buf w_cen;                   // centred f16 path: row / column info words, statistics slots, failure + verdict words
  // slot -> tile tables of the dist / Hamming GEMM launches (hg_dist_kernels.hip: dist_tile_table), a few shapes kept
  struct TileTab {
    uint32_t tiles_m = 0, tiles_n = 0, bm = 0, bn = 0, flags = 0, n_slots = 0;
    uint64_t ref_off = 0, qry_off = 0, used = 0;
    Buf dev;
    std::vector<uint32_t> host;  // (kept: the upload reads it asynchronously)
    hipEvent_t uploaded = nullptr;  // behind the upload: the host copy may be rewritten once it has passed
  } tile_tabs[8];
  uint64_t tile_tab_clock = 0;
  std::string last_kernel_cen; // name of the last centred f16 GEMM queued (it is the DIST kernel when its verdict was positive)
  const void *cen_sig_ref = nullptr, *cen_sig_qry = nullptr;  // operands of the last call that ran on centred f16 operands
  uint32_t cen_sig_r = 0, cen_sig_q = 0, cen_sig_d = 0;
  int last_ham_path = -1;       // last Hamming search: 0 xor + popcount kernel, 1 +-1 byte GEMM (i8), 2 +-1.0 e2m1 GEMM (FP4)
  std::string last_kernel[HG_T_COUNT];  // name of the last kernel launched per timing class (hg_ctx_last_kernel)
  std::string last_kernel_i8;           // ... of the last i8 operand attempt (it is the DIST kernel when the attempt was valid)
  int last_dist_path = -1;      // operand path of the last thresholded dist call: 0 f16 MFMA (raw values), 1 i8 MFMA, 2 integer VALU, 3 f16 MFMA on centred counts
  const void *i8_sig_ref = nullptr, *i8_sig_qry = nullptr;  // operands of the last call that took the i8 path
  uint32_t i8_sig_r = 0, i8_sig_q = 0, i8_sig_d = 0;
  uint32_t i8_skip = 0;         // calls left that skip the i8 attempt after it was vetoed
  Buf w_sorthits; // keys / permutations / scratch of the device-side hit ordering
  // optional per-kernel timing (hg_ctx_enable_timing)
  bool timing = false;
  struct TimedLaunch {
    hipEvent_t e0, e1;
    int cls;
    bool own_e0;  // false: e0 is the previous bracket's e1 (hg_timed chained)
  };
  hipEvent_t t_chain = nullptr;  // closing event of the last bracket while nothing else has been queued behind it
  int t_chain_cls = -1;
  std::vector<TimedLaunch> t_pending;   // recorded, not yet read
  std::vector<hipEvent_t> t_pool;       // reusable events
  // cached batch plan of the last sketch call: when the next call has the same geometry the host
  // neither rebuilds the work-item table nor uploads it again (hg_sketch.h)
  std::shared_ptr<hg_sketch_plan> plan;  // != nullptr: w_gmeta / w_items hold (or are being sent) this plan's tables
  hipEvent_t plan_uploaded = nullptr;    // behind the last upload from the plan staging area (h_plan)
  bool plan_upload_pending = false;
  void *h_plan = nullptr;                // page-locked staging of the plan tables (meta records + work items)
  size_t h_plan_cap = 0;
  // the sync-free sketch step (hg_sketch_step.hip): its check word comes back through h_chk one call late
  hg_sketch_pending pending;
  uint32_t *h_chk = nullptr;   // 2 slots of 16 page-locked words the device writes {flags, ..., seq} into
  uint32_t chk_seq = 0;
  uint64_t n_fast_steps = 0, n_sync_steps = 0, n_redone_steps = 0;  // (hg_ctx_sketch_path_counts: tests, bench)
  // host-fed batches: uploads run on their own stream, one event per sub-batch (hg_sketch_batch)
  hipStream_t copy_stream = nullptr;
  std::vector<hipEvent_t> copy_events;
  // two pinned staging buffers for sub-batches of many small genomes (one packed upload instead of one
  // hipMemcpyAsync per genome); pack_ev[i] marks the last upload that read pack_buf[i]
  void *pack_buf[2] = {nullptr, nullptr};
  size_t pack_cap[2] = {0, 0};  // (sized by need: 2 MB for the 5 Mbp genome of a one-genome call, 66 MB for a batch's sub-batches)
  hipEvent_t pack_ev[2] = {nullptr, nullptr};
  bool pack_used[2] = {false, false};
  // rows [pad_rows, padded rows) of the f16 operand copies are known to be zero (dist tiles hang over)
  const void *pad_a_ptr = nullptr, *pad_b_ptr = nullptr;
  uint32_t pad_a_rows = 0, pad_a_ldk = 0, pad_b_rows = 0, pad_b_ldk = 0;
  std::vector<const void *> lds_attr_done;  // kernels whose dynamic-LDS limit was already raised on this device
  // development / test hooks (hg_ctx_set_debug); never read from the environment
  std::string dbg_dist_tile, dbg_dist_path, dbg_ham_path, dbg_dist_order, dbg_kmer_input, dbg_hostfed, dbg_sketch_path;
  int dbg_sort_buckets = 0;
  uint64_t dbg_pair_limit = 0;  // test hook "pair_limit": pairs one kernel launch of a comparison may enumerate (0: 2^32 - 1, the hit counter's reach)
  // pinned host scratch
  void *h_pin = nullptr;
  size_t h_pin_cap = 0;
  uint32_t *h_res = nullptr;  // 32 words of page-locked host memory the device writes results into (hg_publish_words)
  uint32_t res_seq = 0;       // sequence number of the last publication
};

hg_status hg_fail(hg_ctx *ctx, hg_status s, const std::string &msg);
hg_status hg_ensure(hg_ctx *ctx, hg_ctx::Buf &b, size_t bytes);
hg_status hg_ensure_pinned(hg_ctx *ctx, size_t bytes);
// Result words back to the host without a copy command: a one-wave kernel at the end of the stream's work writes
// n <= 16 words of device memory into the ctx's page-locked result block and raises a sequence number behind them;
// the host polls that word (and the stream's own completion as a fallback) instead of paying a D2H copy command and a
// stream synchronisation.  On return everything queued on the stream before the call has finished.  *out -> the n words.
hg_status hg_publish_words(hg_ctx *ctx, uint32_t *d_words, uint32_t n, const uint32_t **out, uint32_t zero_n = 0);

// RAII bracket: records events around the launches issued while it is alive (no-op unless
// timing is enabled).
// Brackets the launches of one kernel class with events.  after_cls >= 0: the caller queues this bracket's kernels
// directly behind the bracket of class after_cls (nothing in between on the stream) -- its closing event then opens this
// one too: one event record less on the stream (each costs ~5 us of stream time; 19 us per dist call with four of them).
struct hg_timed {
  hg_ctx *c;
  hipEvent_t e0 = nullptr, e1 = nullptr;
  int cls;
  bool own_e0 = true;
  hg_timed(hg_ctx *ctx, int cls_, int after_cls = -1);
  ~hg_timed();
};

#define HG_HIP(ctx, expr)                                                              \
  do {                                                                                 \
    hipError_t e__ = (expr);                                                           \
    if (e__ != hipSuccess)                                                             \
      return hg_fail((ctx), HG_ERR_HIP,                                                \
                     std::string(#expr) + ": " + hipGetErrorString(e__));              \
  } while (0)

// Reads the check word of the ctx's queued sketch step, if there is one, and re-runs the step through the synchronous
// path when it reports a genome that outgrew its hit region or the one-workgroup sort (hg_sketch_step.hip).  Every entry
// point that takes a ctx starts with it (HG_ENTER): whatever it reads or overwrites is final / free by then.
hg_status hg_sketch_resolve(hg_ctx *ctx, bool *redone = nullptr);
#define HG_ENTER(ctx)                                        \
  do {                                                       \
    HG_HIP((ctx), hipSetDevice((ctx)->device));              \
    const hg_status s__ = hg_sketch_resolve((ctx));          \
    if (s__ != HG_OK) return s__;                            \
  } while (0)

// ---- k-mer sampling kernel interface ---------------------------------------------------
// One record per genome of a batch, in device memory.
struct hg_genome_meta {
  uint64_t seq_off;     // byte offset of the genome in d_seq (multiple of 4); packed input: of its 2-bit codes
  uint64_t n_bps;       // length in bytes (bases)
  uint64_t hit_off;     // first slot of the genome's region in the hit buffer
  uint32_t hit_cap;     // slots in that region
  uint32_t item_first;  // index of the genome's first work item
  uint64_t mask_off;    // packed input: byte offset of the genome's not-a-base bitmap in d_seq (hg_pack2: seq_off + padded code bytes)
};
// hg_sketch_batch_dev_packed with the bitmaps at explicit offsets (the streaming path: a genome that came over the link as
// codes + run table has its bitmap rebuilt behind the table, not directly behind the codes)
hg_status hg_sketch_batch_dev_packed_masks(hg_ctx *c, const uint8_t *d_blobs, const uint64_t *code_offs, const uint64_t *mask_offs,
                                           const uint64_t *n_bps, size_t n, const hg_sketch_params *p, int16_t *d_hv,
                                           int32_t *d_norm2, uint32_t *d_nhash);

// starts handled by one work item (one workgroup) of the fast kernel for a given k
uint32_t hg_kmer_item_starts(uint32_t ksize);
// name of the kernel hg_launch_kmer_sample launches for (ksize, canonical), as a profiler prints it
const char *hg_kmer_kernel_name(uint32_t ksize, bool canonical, bool packed);

// Launch the hash + sample kernel over all work items.  d_cnt[g] is incremented once per
// sampled k-mer (it may exceed hit_cap: only the first hit_cap hashes are stored).
// packed: genome g is a hg_pack2 blob at d_seq + seq_off (n_bps = its bases), not ASCII.
// d_group_first (n_groups + 1 entries, or nullptr): workgroup w takes the work items [d_group_first[w], d_group_first[w + 1])
// -- the plan's groups of small genomes (k <= 32; the long-k kernel takes one item per workgroup whatever is passed).
hipError_t hg_launch_kmer_sample(hipStream_t st, const uint8_t *d_seq, const hg_genome_meta *d_meta,
                                 const uint32_t *d_item_genome, uint32_t n_items, uint32_t ksize,
                                 uint64_t threshold, uint64_t seed, bool canonical, uint32_t norm_mode,
                                 uint64_t *d_hits, uint32_t *d_cnt, bool packed = false, const uint32_t *d_group_first = nullptr,
                                 uint32_t n_groups = 0);
// starts per tile / tiles per work item of the k <= 32 kernel (0 for k > 32: no grouping there)
uint32_t hg_kmer_tile_starts(uint32_t ksize);
uint32_t hg_kmer_item_tiles(uint32_t ksize);
// ASCII -> hg_pack2 blobs on the device (bit-identical to the host's hg_pack2): genome i of d_seq at seq_offs[i] with
// lens[i] bases goes to d_blobs + blob_offs[i] (multiples of 16).  d_tab: 3 * n uint64 of device scratch for the tables.
hipError_t hg_launch_pack2(hipStream_t st, const uint8_t *d_seq, const uint64_t *d_tab, uint32_t n, uint32_t blocks_max,
                           uint32_t u2t, uint8_t *d_blobs);

// keys one workgroup can sort in LDS; genomes with more sampled hashes are sorted in place in
// global memory, which needs a power-of-two sized hit region (the host rounds hit_cap up).
#define HG_SORT_LDS_MAX_KEYS 8192u  // keys one workgroup orders in LDS with its counting sort (was 16 384 with the bitonic network behind 8 192)

// ---- sort/unique + encode kernels ---------------------------------------------------------
// Sorts each genome's hits ascending, removes duplicates in place (region start),
// d_ndistinct[g] = distinct count.  max_cnt_pow2 bounds the LDS sort size.
// keys the LDS sort launched for capacity `max_cap` holds per genome (a power of two, at most HG_SORT_LDS_MAX_KEYS);
// with d_todo the launch covers the listed genomes only
uint32_t hg_sort_lds_keys(uint32_t max_cap);
hipError_t hg_launch_sort_unique_todo(hipStream_t st, const hg_genome_meta *d_meta, const uint32_t *d_todo, uint32_t n_todo,
                                      uint64_t *d_hits, const uint32_t *d_cnt, uint32_t *d_ndistinct, uint32_t max_cap,
                                      uint64_t threshold);
// (threshold: every key is below it -- the sampling threshold; it scales the buckets of the counting-sort fast path,
// 0 = bitonic only)
// d_flags != nullptr (the sync-free step): a genome whose raw count exceeds its hit region (HG_STEP_OVERFLOW) or the
// one-workgroup sort (HG_STEP_LARGE_SET) is not sorted -- its bit is or-ed into *d_flags and its distinct count becomes
// HG_NHASH_PENDING, which the encoders skip.
hipError_t hg_launch_sort_unique(hipStream_t st, const hg_genome_meta *d_meta, uint32_t n_genomes,
                                 uint64_t *d_hits, const uint32_t *d_cnt, uint32_t *d_ndistinct,
                                 uint32_t max_cap, uint64_t threshold, uint32_t *d_flags = nullptr);
#define HG_STEP_OVERFLOW 1u
#define HG_STEP_LARGE_SET 2u
// The genomes a sort sized for `done_cap` left out (more raw hits than its LDS held keys), with one sized for max_cap:
// the workgroups find them in the counters themselves -- no list from the host.
hipError_t hg_launch_sort_unique_rest(hipStream_t st, const hg_genome_meta *d_meta, uint32_t n_genomes, uint64_t *d_hits,
                                      const uint32_t *d_cnt, uint32_t *d_ndistinct, uint32_t done_cap, uint32_t max_cap,
                                      uint64_t threshold);
// Last kernel of a sync-free step: d_nhash[g] = d_ndistinct[g], and the step's flag word goes to the page-locked
// check slot with the step's sequence number behind it (h_slot[0] = flags, h_slot[1] = seq).
hipError_t hg_launch_sketch_finish(hipStream_t st, const uint32_t *d_ndistinct, uint32_t *d_nhash, uint32_t n_genomes,
                                   const uint32_t *d_flags, uint32_t *h_slot, uint32_t seq);

// Genomes with more than HG_SORT_LDS_MAX_KEYS sampled hashes: keys are bucketed by value (monotone map, so
// the concatenation of sorted buckets is sorted), every bucket is sorted + de-duplicated in LDS by its own
// workgroup, and the distinct lists are packed back into the genome's hit region.
struct hg_bucket_job {
  uint64_t hit_off;  // the genome's region in the hit buffer (and in the scratch copy)
  uint64_t mul;      // bucket(h) = min(P - 1, mulhi64(h, mul))
  uint32_t n;        // keys (raw hits, duplicates included)
  uint32_t P;        // buckets
  uint32_t bucket_first, chunk_first;  // first bucket / first key chunk of this job in the global lists
  uint32_t genome, pad;
};
#define HG_BUCKET_CHUNK 4096u  // keys per counting / scattering workgroup
// d_bk: 5 * n_buckets + n_jobs uint32 of scratch (zeroed by the launcher); d_fail = d_bk + 5 * n_buckets gets
// 1 for every job whose buckets could not be de-duplicated in LDS (the caller then runs the in-place sort).
hipError_t hg_launch_sort_large(hipStream_t st, const hg_bucket_job *d_jobs, uint32_t n_jobs,
                                const uint32_t *d_chunk_job, uint32_t n_chunks, const uint32_t *d_bucket_job,
                                uint32_t n_buckets, uint32_t *d_bk, uint64_t *d_hits, uint64_t *d_tmp,
                                uint32_t *d_ndistinct, uint32_t bucket_cap_keys);
// in-place global-memory sort + unique of the listed genomes (power-of-two sized hit regions)
hipError_t hg_launch_sort_inplace(hipStream_t st, const hg_genome_meta *d_meta, const uint32_t *d_todo,
                                  uint32_t n_todo, uint64_t *d_hits, const uint32_t *d_cnt, uint32_t *d_ndistinct);

// Genomes with more than HG_ENC_SLAB distinct hashes can be encoded by several workgroups: d_items[i] =
// {genome, slab | slot << 16} for every slab of HG_ENC_SLAB hashes (planned from an upper bound of the distinct
// count), d_genomes[slot] = genome, d_accum = n_genomes * hv_d uint32 of scratch.  Slabs and slots are < 65536.
#define HG_ENC_SLAB 32768u
#define HG_ENC_WAVE_MAX 16368u  // hashes one wave encodes without leaving its bit-sliced counters (16 * 1023)
struct hg_encode_split {
  const uint32_t *d_items;    // uint2 pairs
  const uint32_t *d_genomes;
  uint32_t *d_accum;
  uint32_t n_items, n_genomes;
};
hipError_t hg_launch_encode(hipStream_t st, const hg_genome_meta *d_meta, uint32_t n_genomes,
                            const uint64_t *d_hits, const uint32_t *d_ndistinct, uint32_t hv_d,
                            uint32_t layout, int16_t *d_hv, int32_t *d_norm2, const hg_encode_split *split = nullptr,
                            uint32_t max_hashes = ~0u /* upper bound of the distinct counts, if the host knows one */);

// ---- dist kernels ------------------------------------------------------------------------------
struct hg_dist_args {
  const int16_t *ref_hv;
  const int32_t *ref_n2;
  const int16_t *qry_hv;
  const int32_t *qry_n2;
  uint32_t R, Q, hv_d, ksize;
  float *ani_out;        // full matrix or nullptr
  hg_ani_hit *hits;      // thresholded output or nullptr
  uint32_t *hit_count;   // device counter
  uint32_t hit_cap;
  float ani_th;
  int symmetric;
  uint32_t ref_off = 0, qry_off = 0;  // global indices of row 0 / column 0 when the call is a block of a larger matrix
  // the reference side as prepared byte operands (hg_dist_prep_ops_dev on the GPUs that own the rows, gathered by the caller):
  // ref_hv is unused then.  ref_ops: hg_dist_ops_padded_rows(R) rows of hg_dist_ops_row_bytes(hv_d); ref_meta: R records of
  // hg_dist_ops_meta_bytes(); ref_flags: the owners' n_flags failure words; ref_index: optional global index per row
  const uint8_t *ref_ops = nullptr;
  const void *ref_meta = nullptr;
  const uint32_t *ref_flags = nullptr;
  uint32_t n_flags = 0;
  const uint32_t *ref_index = nullptr;
};
size_t hg_dist_ops_row_bytes_impl(uint32_t hv_d);
size_t hg_dist_ops_meta_bytes_impl();
size_t hg_dist_ops_padded_rows_impl(size_t n);
hg_status hg_run_dist_prep_ops(hg_ctx *c, const int16_t *d_hv, uint32_t rows, uint32_t hv_d, uint8_t *d_ops, void *d_meta,
                               uint32_t *d_flag);
// d_verdict (two uint32: code, window length) != nullptr allows the speculative schedule: prepass, on-device
// exactness verdict (0 = one f32 window covers K; 1 / 2 = windows of 2 048 / 1 024 dims; 3 = neither) and the
// GEMM launches guarded by it are queued without a host round trip.  *speculated = -1 if the call was not
// speculative, else the highest verdict code a guarded launch covers: the caller reads the verdict back with
// its own results and calls again without d_verdict if it is larger.
hg_status hg_run_dist(hg_ctx *ctx, const hg_dist_args &a, uint32_t *d_verdict = nullptr, int *speculated = nullptr);

// Bit-packed Hamming search as an exact GEMM on the matrix pipe (+-1 operands expanded from the bits -- e2m1 nibbles
// for v_mfma_scale_f32_16x16x128_f8f6f4 or bytes for v_mfma_i32_16x16x64_i8; the ANI GEMM's tiles, LDS-DMA staging and
// hit lists): all pairs with distance <= max_dist appended to d_hits through *d_count (zeroed by the caller).
hg_status hg_run_hamming_mfma(hg_ctx *c, const uint32_t *d_ref_bits, uint32_t R, const uint32_t *d_qry_bits, uint32_t Q,
                              uint32_t hv_d, uint32_t max_dist, hg_ham_hit *d_hits, uint32_t *d_count, uint32_t cap,
                              uint32_t ref_off, uint32_t qry_off, int path /* 1 = +-1 bytes (i8), 2 = e2m1 nibbles (FP4) */);
