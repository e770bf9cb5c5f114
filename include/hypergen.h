/*
 * hypergen.h -- C ABI of libhypergen_hip.so, the MI355X (gfx950) implementation of
 * the HyperGen sketch + ANI hot path.
 *
 * This is the drop-in boundary.  Each entry point names the reference interface
 * (wh-xu/Hyper-Gen, file:line) it replaces; INTEGRATION.md shows the Rust
 * `extern "C"` block a maintainer would add in src/sketch_cuda.rs / src/dist.rs.
 *
 * Conventions
 *   - plain C types only; every call returns an hg_status (the reference
 *     .unwrap()s and aborts, src/sketch_cuda.rs:134-156 -- here the caller decides);
 *   - `hg_ctx` owns one device, one stream and growable device workspaces; a ctx is
 *     NOT thread-safe, create one per host thread (the reference binds one shared
 *     CudaDevice per rayon worker, src/sketch_cuda.rs:82);
 *   - "host" entry points take host pointers and stage through the ctx;
 *     "_dev" entry points take device pointers (HBM resident, no PCIe traffic);
 *   - outputs are caller allocated with explicit capacities;
 *   - there is NO CPU fallback: without a usable HIP device hg_ctx_create fails.
 */
#ifndef HYPERGEN_H
#define HYPERGEN_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef struct hg_ctx hg_ctx;

typedef enum {
  HG_OK = 0,
  HG_ERR_INVALID = 1,     /* bad argument                                          */
  HG_ERR_NO_DEVICE = 2,   /* no HIP device / device id out of range                */
  HG_ERR_HIP = 3,         /* a HIP runtime call failed (see hg_last_error)         */
  HG_ERR_OOM = 4,         /* device or host allocation failed                      */
  HG_ERR_CAPACITY = 5,    /* caller's output buffer too small; *n_out = needed     */
  HG_ERR_UNSUPPORTED = 6, /* parameter combination not implemented on the device   */
  HG_ERR_IO = 7,          /* file could not be read / written / parsed             */
  HG_ERR_INEXACT = 8      /* integer dot products cannot be evaluated exactly      */
} hg_status;

const char *hg_status_str(hg_status s);
/* message of the last failing call on this ctx (or on ctx creation if ctx == NULL) */
const char *hg_last_error(const hg_ctx *ctx);
/* library version string */
const char *hg_version(void);

/* ---- context ------------------------------------------------------------------
 * replaces CudaDevice::new(0) + load_ptx(..) + bind_to_thread()
 * (src/sketch_cuda.rs:52-60,82).  Kernels are linked into the library, there is no
 * PTX/code-object loading step. */
hg_status hg_ctx_create(int device_id, hg_ctx **out);
void hg_ctx_destroy(hg_ctx *ctx);
/* run all subsequent work on an existing hipStream_t (e.g. torch's current stream).  The handle is used as
 * given: NULL is HIP's default (null) stream -- which IS torch's current stream unless the caller changed
 * it -- so that the ctx's kernels are ordered with the caller's own work on that stream.
 * hg_ctx_reset_stream goes back to the ctx's private non-blocking stream (the state after hg_ctx_create).
 * Both first wait for the work already queued on the stream being left (the ctx workspaces are ordered by
 * one stream at a time). */
hg_status hg_ctx_set_stream(hg_ctx *ctx, void *hip_stream);
hg_status hg_ctx_reset_stream(hg_ctx *ctx);
hg_status hg_ctx_sync(hg_ctx *ctx);
/* How many sketch steps (hg_sketch_batch_dev / _packed calls, sub-batches of the host-fed entry points) the ctx has queued
 * without a host round trip, how many it ran through the synchronous path, and how many of the former it had to run again
 * because their check word asked for it (see "Completion of hg_sketch_batch_dev" below).  NULL pointers are skipped. */
hg_status hg_ctx_sketch_step_counts(hg_ctx *ctx, uint64_t *sync_free, uint64_t *synchronous, uint64_t *redone);
int hg_device_count(void);
/* development / test hook (no reference counterpart): force an internal code path of THIS ctx.
 * keys: "dist_tile" = "" | "small" | "big" | "wide"      (GEMM tile geometry)
 *       "dist_path" = "" | "f16" | "i8" | "cen"            (operand format of the ANI GEMM: raw values as f16 only / try byte
 *                                                           operands / try centred counts as f16 -- also on small problems)
 *       "dist_order" = "" | "plain" | "legacy"             ("plain": a self-comparison does not run its diagonal tiles first;
 *                                                           "legacy": workgroups derive their tile from blockIdx instead of
 *                                                           reading the host's balanced slot -> tile table)
 *       "ham_path"  = "" | "popc" | "mfma" | "fp4"         (Hamming search: xor + popcount, +-1 byte GEMM, +-1 e2m1 GEMM)
 *       "kmer_input" = "" | "packed"                       ("packed": batches that arrive as ASCII are 2-bit packed on the
 *                                                           device first and take the packed-input kernels)
 *       "hostfed" = "" | "ascii" | "packed"                (what hg_sketch_batch / hg_kmer_hash_sample send over the link:
 *                                                           the library's choice / always ASCII / always 2-bit packed on the host)
 *       "sort_test_buckets" = "<n>"   (bucket count of the large-set sort; 0 = automatic)
 *       "pair_limit" = "<n>"          (pairs one launch of a thresholded comparison / search may enumerate before the call is
 *                                      split into blocks of reference rows; 0 = 2^32 - 1, the reach of the hit counter)
 * Nothing in the library reads environment variables. */
hg_status hg_ctx_set_debug(hg_ctx *ctx, const char *key, const char *value);

/* per-kernel device timing: when enabled, every kernel launch of the sketch / dist entry
 * points is bracketed by HIP events on the ctx's stream.  hg_ctx_timings waits for the last
 * bracket and returns, per kernel class, the SUM of the launch durations (ms) and the number
 * of launches since the previous hg_ctx_timings call. */
#define HG_T_KMER 0      /* k-mer hash + sample            */
#define HG_T_SORT 1      /* sort + unique                  */
#define HG_T_ENCODE 2    /* HV encode + norm               */
#define HG_T_DIST_PREP 3 /* i16 -> f16 + exactness bounds  */
#define HG_T_DIST 4      /* ANI GEMM (MFMA or integer)     */
#define HG_T_COUNT 5
hg_status hg_ctx_enable_timing(hg_ctx *ctx, int on);
hg_status hg_ctx_timings(hg_ctx *ctx, float ms_sum[HG_T_COUNT], uint32_t launches[HG_T_COUNT]);
/* name of the kernel the last call launched for timing class `cls` (HG_T_KMER, HG_T_DIST), spelled as rocprofv3
 * prints it ("kmer_sample_shared<21, true, false>", "dist_mfma_kernel<false, false, true, true, 5, true, false, false, false>");
 * "" if none.  A measurement harness uses it to check that a committed profile belongs to the kernel that ran. */
const char *hg_ctx_last_kernel(const hg_ctx *ctx, int cls);

/* minimal device-memory helpers for callers that have no HIP binding of their own
 * (cudarc's htod_copy / alloc_zeros / sync_reclaim, src/sketch_cuda.rs:134,138,156) */
hg_status hg_dev_alloc(hg_ctx *ctx, size_t bytes, void **dptr);
hg_status hg_dev_free(hg_ctx *ctx, void *dptr);
hg_status hg_copy_h2d(hg_ctx *ctx, void *dst_dev, const void *src_host, size_t bytes);
hg_status hg_copy_d2h(hg_ctx *ctx, void *dst_host, const void *src_dev, size_t bytes);

/* ---- parameters --------------------------------------------------------------- */
#define HG_LAYOUT_SCALAR 0u /* encode_hash_hd       src/hd.rs:94-112               */
#define HG_LAYOUT_AVX2 1u   /* encode_hash_hd_avx2  src/hd.rs:14-92 (x86-64 output) */
#define HG_NORM_ACGT 0u     /* only ACGTacgt are bases (src/cuda_kernel.cu:277-296) */
#define HG_NORM_U2T 1u      /* additionally u/U -> T (needletail normalize)         */

/* mirrors SketchParams (src/types.rs:83-113); defaults k=21 scaled=1500 seed=123
 * canonical=1 hv_d=4096 */
typedef struct {
  uint32_t ksize;     /* 1..255 (u8 in the reference): 1..32 kmer_sample_shared, 33..64 kmer_sample_long<k>, 65..255 run-time k */
  uint32_t canonical; /* 0/1 (honoured like src/cuda_kernel.cu:306-314)             */
  uint64_t scaled;    /* threshold = UINT64_MAX / scaled (src/sketch.rs:73)          */
  uint64_t seed;
  uint32_t hv_d;      /* HV dimension; chunks of 64 are filled (src/hd.rs:34,102)    */
  uint32_t hv_layout; /* HG_LAYOUT_*                                                 */
  uint32_t norm_mode; /* HG_NORM_*                                                   */
  uint32_t reserved;
} hg_sketch_params;

void hg_sketch_params_default(hg_sketch_params *p);

/* ---- k-mer hash + FracMinHash sample --------------------------------------------
 * replaces extract_kmer_t1ha2_cuda + kernel cuda_kmer_t1ha2
 * (src/sketch_cuda.rs:120-166, src/cuda_kernel.cu:250-321; CPU twin
 * extract_kmer_hash, src/sketch.rs:71-98).
 * seq   : read_merge_seq layout (src/fastx_reader.rs:6-29), host memory.
 * output: the DISTINCT sampled hashes, ascending (HashSet semantics, lossless: no
 *         per-thread slot cap, hash value 0 is kept).
 * If more than `cap` distinct hashes exist: HG_ERR_CAPACITY and *n_out = count. */
hg_status hg_kmer_hash_sample(hg_ctx *ctx, const uint8_t *seq, size_t n_bps,
                              uint32_t ksize, uint64_t threshold, uint64_t seed,
                              int canonical, uint32_t norm_mode, uint64_t *out_hashes,
                              size_t cap, size_t *n_out);

/* ---- HV encode -------------------------------------------------------------------
 * replaces hd::encode_hash_hd{,_avx2} + dist::compute_hv_l2_norm
 * (src/hd.rs:14-112, src/dist.rs:132-137).  `hashes` must be distinct (host). */
hg_status hg_hv_encode(hg_ctx *ctx, const uint64_t *hashes, size_t n, uint32_t hv_d,
                       uint32_t hv_layout, int16_t *hv_out, int32_t *norm2_out);

/* ---- whole-batch sketch ------------------------------------------------------------
 * what the rayon loop body of src/sketch.rs:35-48 / src/sketch_cuda.rs:79-96 does per
 * file (hash+sample -> set -> HV encode -> norm), for n genomes in one call.
 *
 * _dev: d_seq holds all genomes (device memory); genome i occupies
 * [offsets[i], offsets[i] + lens[i]) (host arrays).  offsets must be multiples of 4
 * and the allocation must be readable for 32 bytes past every genome's end.
 * d_hv      : n * hv_d int16 (device), d_norm2 : n int32 (device),
 * d_nhash   : n uint32 (device) = number of distinct sampled hashes. */
hg_status hg_sketch_batch_dev(hg_ctx *ctx, const uint8_t *d_seq, const uint64_t *offsets,
                              const uint64_t *lens, size_t n, const hg_sketch_params *p,
                              int16_t *d_hv, int32_t *d_norm2, uint32_t *d_nhash);
/* Completion of hg_sketch_batch_dev / _packed.  The call queues the whole step -- hash + sample, sort / unique, encode --
 * on the ctx's stream and returns; no host round trip separates the kernels.  What the host used to read back between
 * them (did a genome outgrow its hit region -- more than twice its expected number of sampled k-mers + 1 024, the
 * signature of a sampled k-mer repeated thousands of times -- or the one-workgroup sort) comes back as ONE check word
 * behind the last kernel, and the library reads it when the NEXT call on the ctx, or hg_ctx_sync, arrives: a step that
 * reported such a genome is then run again through the synchronous path (exact capacities, multi-workgroup sort), before
 * that call does anything else.  Hence:
 *   - d_seq, d_hv, d_norm2, d_nhash belong to the library until the next call on the ctx or hg_ctx_sync has returned;
 *   - results are FINAL after hg_ctx_sync (or any later call on the ctx) has returned.  A consumer that orders itself
 *     behind the call on the stream alone sees final rows too, except for the genomes of such a step: their d_nhash holds
 *     HG_NHASH_PENDING (and their HV rows are untouched) until the library has redone the step;
 *   - an error of the re-run is returned by the call that triggered it.
 * Batches whose genomes are EXPECTED to exceed the one-workgroup sort (more than ~7 000 sampled k-mers: 10 Mbp at
 * scaled = 1 500) take the synchronous path at once and are final in stream order; hg_ctx_set_debug(ctx, "sketch_path",
 * "sync") sends every batch that way (the behaviour up to round 5: counters read back between sort and encode). */
#define HG_NHASH_PENDING 0xFFFFFFFFu
/* How the library would lay a batch out for the k-mer launch (host arithmetic only; no ctx, no device): counts[0] = work items
 * (pieces of 27 432 k-mer starts for k <= 21, 27 324 for k <= 32, 12 288 beyond), [1] = workgroups -- for k <= 32 the work
 * items of consecutive SMALL genomes share a workgroup, up to three full items' worth of tiles --, [2] = hit slots, [3] = largest
 * hit region, [4] = largest expected sampled count of a genome, [5] = tiles of a full work item (0 for k > 32).  group_first
 * (optional, cap entries): first work item of every workgroup, then the item count (counts[1] + 1 entries; k <= 32 only). */
hg_status hg_sketch_plan_describe(const uint64_t *offsets, const uint64_t *lens, size_t n, uint32_t ksize, uint64_t scaled,
                                  uint64_t counts[6], uint32_t *group_first, size_t cap);
/* The same batch with the genomes resident as 2-bit PACKED bases: genome i is the hg_pack2 blob (layout below:
 * 4 bases per byte + the not-a-base bitmap, 0.375 bytes per base in HBM instead of 1) at d_blobs + offsets[i]
 * (multiples of 16; 32 readable bytes behind every blob), n_bps[i] = its number of bases.  The k-mer kernels read the
 * codes as they lie -- no ASCII copy exists on the device -- and the result is bit-identical to hg_sketch_batch_dev on
 * the sequences the blobs were packed from (under the norm_mode they were packed with; p->norm_mode is not applied
 * again).  The pattern of the reference's second kernel (src/cuda_kernel.cu:15-69 on NT4 codes, src/sketch_cuda.rs:23-32). */
hg_status hg_sketch_batch_dev_packed(hg_ctx *ctx, const uint8_t *d_blobs, const uint64_t *offsets,
                                     const uint64_t *n_bps, size_t n, const hg_sketch_params *p,
                                     int16_t *d_hv, int32_t *d_norm2, uint32_t *d_nhash);
/* host buffers in, host results out.  The sequences cross the link in sub-batches that upload while the previous one is
 * sketched.  What crosses is the library's choice and never changes a result bit: a batch of >= 32 MB on a host with >= 4
 * usable cores is 2-bit packed by up to 16 host threads of the call (hg_pack2 blobs, 0.375 bytes per base, in pieces of
 * 1 Mbase into page-locked staging; the packed-input kernels sketch them) -- 2.3x the ASCII rate on the measured box,
 * where host DRAM (114 GB/s of packing reads) then limits instead of PCIe; the first sub-batch is timed and the rest goes
 * as ASCII when the host packs slower than the link carries.  A call with n = 1 packs on the calling thread when the
 * source is pageable memory, or when >= 3 other host-fed calls of the process are in flight (the reference's
 * one-call-per-genome pattern from a thread pool, src/sketch_cuda.rs:79-96: the calls share one link) and the calling
 * threads together pack fast enough for that to pay -- their rate is measured in the packed calls; when it falls short
 * (a host whose memory is busy) the following calls go as ASCII for a while before packing is tried again.  Debug key
 * "hostfed" = "ascii" | "packed" pins the choice.  The same rule applies to hg_kmer_hash_sample. */
hg_status hg_sketch_batch(hg_ctx *ctx, const uint8_t *const *seqs, const size_t *lens,
                          size_t n, const hg_sketch_params *p, int16_t *hv_out,
                          int32_t *norm2_out, uint32_t *nhash_out);

/* ---- ANI ------------------------------------------------------------------------------
 * replaces dist::compute_hv_ani / compute_pairwise_ani (src/dist.rs:139-161,231-294).
 * ref_hv: R x hv_d int16 row-major (decompressed sketches), qry_hv: Q x hv_d.
 * ani_out[r * Q + q] = ANI in percent (0..100), float32 arithmetic in the reference's
 * operation order, INCLUDING its logarithm: Rust's f32::ln is the C library's logf, and the device evaluates glibc's logf
 * algorithm (sysdeps/ieee754/flt-32/e_logf.c) operation for operation (csrc/hg_logf.h), so an ANI value is the float the
 * reference computes for the same dot product and norms, bit for bit -- and with it the third decimal of a TSV line, the
 * side of `ani >= ani_th` and the order of near-ties.  Dot products are exact integers (HG_ERR_INEXACT is never a silent
 * rounding: it is returned only if no exact device path applies). */
hg_status hg_dist_full(hg_ctx *ctx, const int16_t *ref_hv, const int32_t *ref_norm2,
                       size_t R, const int16_t *qry_hv, const int32_t *qry_norm2, size_t Q,
                       uint32_t hv_d, uint32_t ksize, float *ani_out);
hg_status hg_dist_full_dev(hg_ctx *ctx, const int16_t *d_ref_hv, const int32_t *d_ref_norm2,
                           size_t R, const int16_t *d_qry_hv, const int32_t *d_qry_norm2,
                           size_t Q, uint32_t hv_d, uint32_t ksize, float *d_ani_out);
/* Completion of the *_dev entry points: their device inputs are read, and their device outputs written,
 * in stream order on the ctx's stream (hg_ctx_set_stream), so work the caller queues on that stream before
 * / after the call is ordered with it.  When hg_dist_dev / hg_hamming_search_dev (and their _block forms) return, every
 * kernel queued on the stream before and by the call has finished (the hit count comes back through a page-locked
 * block the last kernel writes and the host polls for a bounded time before it falls back to a blocking stream
 * synchronisation); one 64-byte hipMemsetAsync that re-zeroes the call's counter words may still be pending on the
 * stream -- it touches nothing the caller sees.  hg_dist_full_dev may return with its last kernels still running --
 * hg_ctx_sync (or the caller's own stream synchronisation) completes them; for hg_sketch_batch_dev see "Completion of
 * hg_sketch_batch_dev" above. */
/* The ANI formula's two pieces on their own (diagnostics): d_out[i] = the library's logf of d_x[i] -- or, with d_x ==
 * NULL, of the float whose bit pattern is first_bits + i (an exhaustive sweep needs no input array); d_ani[i] = the tail
 * of compute_pairwise_ani (src/dist.rs:153-160) for the integer dot product d_dot[i] and the norms.  Stream-ordered. */
hg_status hg_logf_dev(hg_ctx *ctx, const float *d_x, uint32_t first_bits, size_t n, float *d_out);
hg_status hg_ani_from_dots_dev(hg_ctx *ctx, const int32_t *d_dot, const int32_t *d_norm2_r, const int32_t *d_norm2_q,
                               size_t n, uint32_t ksize, float *d_ani);

/* one reported pair: what dump_ani_file prints per line (src/utils.rs:275-285) */
typedef struct {
  uint32_t ref_idx;
  uint32_t qry_idx;
  float ani;
} hg_ani_hit;

/* thresholded form: pairs with ani >= ani_th.  symmetric != 0 enumerates only
 * ref_idx < qry_idx (src/dist.rs:243-265).  Hits come back in no particular order
 * (hg_sort_ani_hits gives the reference's file order).  *n_out = number of hits found;
 * if it exceeds cap only cap are stored and HG_ERR_CAPACITY is returned.
 * `out` is host memory for hg_dist and device memory for hg_dist_dev.
 * A set compared with itself (the reference's path_r == path_q case, src/dist.rs:13) is recognised by identical
 * pointers and row counts on both sides: hg_dist uploads it once, and the device path prepares its operands once
 * and runs the tiles on the matrix diagonal -- where such a comparison has its dense blocks of hits -- first.  The
 * result is the same set of hits either way.
 * The kernels count hits in 32 bits: a call of more than 2^32 - 1 pairs (66 000 x 66 000 and up) runs as blocks of
 * reference rows of fewer pairs each (hg_dist_block_dev does that by itself); counts add up in 64 bits, and once `out` is
 * full the remaining blocks only count.  hg_hamming_search(_block)_dev likewise. */
hg_status hg_dist(hg_ctx *ctx, const int16_t *ref_hv, const int32_t *ref_norm2, size_t R,
                  const int16_t *qry_hv, const int32_t *qry_norm2, size_t Q, uint32_t hv_d,
                  uint32_t ksize, int symmetric, float ani_th, hg_ani_hit *out, size_t cap,
                  size_t *n_out);
hg_status hg_dist_dev(hg_ctx *ctx, const int16_t *d_ref_hv, const int32_t *d_ref_norm2,
                      size_t R, const int16_t *d_qry_hv, const int32_t *d_qry_norm2, size_t Q,
                      uint32_t hv_d, uint32_t ksize, int symmetric, float ani_th,
                      hg_ani_hit *d_out, size_t cap, size_t *n_out);

/* One block of a larger R x Q matrix (multi-GPU decomposition): rows ref_off.. and columns
 * qry_off.. of the global enumeration.  Hits carry GLOBAL indices and `symmetric` keeps global ref < global qry,
 * so the union over the blocks of a partition equals the one-call result. */
hg_status hg_dist_block_dev(hg_ctx *ctx, const int16_t *d_ref_hv, const int32_t *d_ref_norm2, size_t R,
                            size_t ref_off, const int16_t *d_qry_hv, const int32_t *d_qry_norm2, size_t Q,
                            size_t qry_off, uint32_t hv_d, uint32_t ksize, int symmetric, float ani_th,
                            hg_ani_hit *d_out, size_t cap, size_t *n_out);

/* ---- sharded dist: reference operands prepared where the rows live -----------------------------------
 * hg_dist_block_dev needs every reference row as i16 (8 KiB at D = 4096) and converts ALL of them to its byte operands
 * itself: with the references spread over N GPUs that is an exchange of R x D x 2 bytes and an N-fold repeated prepass.
 * Here the rank that OWNS a row prepares it once -- hg_dist_prep_ops_dev: the centred byte operand (src/hd.rs:29,84-87:
 * hv = 2 * count - n, so (hv + e) >> 1 fits a byte), 4 KiB + 128 at D = 4096, and a 72-byte control record -- and the
 * exchange moves those (half the bytes; the reference's i16 rows never travel).  hg_dist_block_ops_dev is
 * hg_dist_block_dev with the reference side given that way; the query side is still the caller's own i16 rows.
 *   d_ops   : rows x hg_dist_ops_row_bytes(hv_d) bytes;  d_meta : rows x hg_dist_ops_meta_bytes() bytes (opaque);
 *   d_flag  : ONE device word per prepared block: 0 = every row fits the byte scheme.  The consumer passes the flag
 *             words of all blocks it gathered (d_flags, n_flags).
 *   d_ref_ops (consumer): room for hg_dist_ops_padded_rows(R) rows -- the tiles overhang, the call zeroes the rows behind R.
 *   d_ref_index: NULL (row i is global reference ref_off + i) or the global index of every row, for a gathered block
 *             whose rows are not one contiguous range (chunked exchanges); not with `symmetric`.
 * Returns HG_ERR_INEXACT, with nothing reported, when an owner's flag is set or the call's own query rows do not fit
 * the scheme (sketches beyond ~6 000 hashes at D = 4096): the caller then gathers the i16 rows and uses
 * hg_dist_block_dev -- same hits either way.  hv_d <= 8192, hv_d % 8 == 0. */
size_t hg_dist_ops_row_bytes(uint32_t hv_d);
size_t hg_dist_ops_meta_bytes(void);
size_t hg_dist_ops_padded_rows(size_t rows);
hg_status hg_dist_prep_ops_dev(hg_ctx *ctx, const int16_t *d_hv, size_t rows, uint32_t hv_d, uint8_t *d_ops,
                               uint8_t *d_meta, uint32_t *d_flag);
hg_status hg_dist_block_ops_dev(hg_ctx *ctx, const uint8_t *d_ref_ops, const uint8_t *d_ref_meta, const int32_t *d_ref_norm2,
                                size_t R, size_t ref_off, const uint32_t *d_ref_index, const uint32_t *d_flags,
                                size_t n_flags, const int16_t *d_qry_hv, const int32_t *d_qry_norm2, size_t Q,
                                size_t qry_off, uint32_t hv_d, uint32_t ksize, int symmetric, float ani_th,
                                hg_ani_hit *d_out, size_t cap, size_t *n_out);

/* Which exact operand path the last hg_dist / hg_dist_dev / hg_dist_block_dev call of this ctx took (all give the same
 * integers): 0 = the raw values as f16 operands on v_mfma_f32_16x16x32_f16 (exact f32 windows), 1 = centred i8 operands on
 * v_mfma_i32_16x16x64_i8 (sketches of up to ~6 000 hashes at D = 4096; decided on the device), 2 = integer VALU fallback,
 * 3 = the centred counts as f16 operands (one exact f32 window up to ~16 000 hashes per sketch at D = 4096: where byte
 * operands stop); -1 = none yet. */
int hg_ctx_last_dist_path(const hg_ctx *ctx);

/* order of dump_ani_file (src/utils.rs:262-269): stable ascending sort by ANI over the
 * reference's pair enumeration, then reversed.  Host side. */
void hg_sort_ani_hits(hg_ani_hit *hits, size_t n, size_t Q, int symmetric);

/* the same order produced on the device (stable radix passes over the enumeration key, then the ANI), in place on
 * a device-resident hit list; stream-ordered, returns without synchronising.  _staged: host list in and out through
 * the ctx (upload, device sort, download) -- a 10^6-hit list orders in ~1 ms instead of ~100 ms of host sort. */
hg_status hg_sort_ani_hits_dev(hg_ctx *ctx, hg_ani_hit *d_hits, size_t n, size_t Q);
hg_status hg_sort_ani_hits_staged(hg_ctx *ctx, hg_ani_hit *hits, size_t n, size_t Q);

/* `search` (an empty stub in the reference, src/main.rs:22-24; defined here): per query the k best references of
 * a hit list (e.g. the output of hg_dist_dev), descending ANI, ties by ascending reference index.
 * d_out: Q * k entries, query q's results at d_out[q*k ..], unused slots have ref_idx = 0xFFFFFFFF;
 * d_counts[q] = number of valid entries (<= k).  Device pointers; stream-ordered. */
hg_status hg_topk_per_query_dev(hg_ctx *ctx, const hg_ani_hit *d_hits, size_t n, size_t Q, uint32_t k,
                                hg_ani_hit *d_out, uint32_t *d_counts);

/* ---- sketch compression (host side; src/hd.rs:114-232) -------------------------------- */
uint32_t hg_hv_quant_bits(const int16_t *hv, uint32_t hv_d);
/* packed holds hg_hv_packed_bytes(hv_d, quant_bits) = quant_bits * (hv_d >> 3) bytes (src/hd.rs:146).  Like the
 * reference, only whole blocks of 256 dimensions are packed (src/hd.rs:147): for hv_d % 256 != 0 the bytes behind
 * them are zero and hg_hv_unpack gives every dimension behind the last whole block the value -2^(quant_bits-1)
 * (src/hd.rs:194,206-212) -- the round trip is lossless only for multiples of 256.  quant_bits = 16 reproduces the
 * reference's sign-extension spill (src/hd.rs:140-141) bit for bit. */
size_t hg_hv_packed_bytes(uint32_t hv_d, uint32_t quant_bits);
hg_status hg_hv_pack(const int16_t *hv, uint32_t hv_d, uint32_t quant_bits, uint8_t *packed);
hg_status hg_hv_unpack(const uint8_t *packed, uint32_t hv_d, uint32_t quant_bits, int16_t *hv);
/* The OTHER payload layout: what the reference writes and reads on a host WITHOUT AVX2 (src/hd.rs:158-166, 213-231).
 * (quant_bits * hv_d + 16) / 16 int16 words -- always one word more than the bits need when that product is a multiple
 * of 16 -- holding the low quant_bits bits of every value LSB first, NO offset added.  Decoding follows the reference
 * to the letter: `if v > (1 << (q-1)) { v - (1 << q) }` in i16 -- strictly greater, so -2^(q-1) (low bits 100..0) comes
 * back as +2^(q-1), and at quant_bits = 16 (`1 << 15` = -32768, `1 << 16` wraps to 1) every value but -32768 comes back
 * one lower: the reference's naive round trip is lossy exactly there, and so is this one.
 * The two layouts never have the same payload length (the naive one is longer by at least one word), which is how
 * hg_hv_payload_layout -- and through it `hyper-gen dist` / `search` -- tells them apart: HG_PAYLOAD_BITPACKER8X,
 * HG_PAYLOAD_NAIVE, or -1 for a length that is neither.  `hyper-gen sketch` writes the naive layout only behind
 * `--pack_layout naive` (an extension flag; the default is what the reference writes on every AVX2 host). */
enum { HG_PAYLOAD_BITPACKER8X = 0, HG_PAYLOAD_NAIVE = 1 };
size_t hg_hv_packed_bytes_naive(uint32_t hv_d, uint32_t quant_bits);
hg_status hg_hv_pack_naive(const int16_t *hv, uint32_t hv_d, uint32_t quant_bits, uint8_t *packed);
hg_status hg_hv_unpack_naive(const uint8_t *packed, uint32_t hv_d, uint32_t quant_bits, int16_t *hv);
int hg_hv_payload_layout(uint32_t hv_d, uint32_t quant_bits, size_t payload_bytes);
/* decompress_file_sketch (src/hd.rs:171-180: one rayon task per sketch) on the device: n payloads, payload i at
 * d_payloads + offsets[i] (byte offsets of any alignment into a device buffer of payloads_bytes bytes -- e.g. the image of
 * a .sketch file, see hg_sketch_file_payload_offset; host array) in layout layouts[i] with quant_bits[i] (host arrays;
 * layouts == NULL: all BitPacker8x), decoded to row i of the n x hv_d int16 matrix d_hv -- the same integers as
 * hg_hv_unpack / hg_hv_unpack_naive, rows and dimensions behind the last whole block included.  The payloads are 4.6 KB
 * per sketch at quant_bits = 9 against 8 KB of int16: `hyper-gen dist` uploads the file's payload bytes as they are and
 * decodes them here instead of unpacking on host threads and uploading the matrix.  Stream-ordered; the host arrays are
 * consumed before the call returns.  (hg_hv_unpack_kernel) */
hg_status hg_hv_unpack_batch_dev(hg_ctx *ctx, const uint8_t *d_payloads, size_t payloads_bytes, const uint64_t *offsets,
                                 const uint8_t *quant_bits, const uint8_t *layouts, size_t n, uint32_t hv_d, int16_t *d_hv);

/* ---- .sketch files (host side; src/types.rs:224-235, src/utils.rs:234-258) ------------ */
typedef struct {
  uint8_t ksize;
  uint8_t canonical;
  uint8_t hv_quant_bits;
  uint8_t pad;
  int32_t hv_norm_2;
  uint64_t scaled;
  uint64_t seed;
  uint64_t hv_d;
  const char *file_str; /* UTF-8, NUL terminated                                      */
  const int16_t *hv;    /* payload exactly as stored (packed when hv_quant_bits < 16) */
  uint64_t hv_len;      /* number of int16 in the payload                             */
} hg_file_sketch;

typedef struct hg_sketch_file hg_sketch_file; /* owns the records of a loaded file */

hg_status hg_sketch_file_write(const char *path, const hg_file_sketch *recs, size_t n);
hg_status hg_sketch_file_read(const char *path, hg_sketch_file **out);
/* The same parse without the payload copies: the records' `hv` are NULL (hv_len is set), the file's bytes stay in one
 * buffer (hg_sketch_file_image) and record i's payload starts at hg_sketch_file_payload_offset(f, i) in it -- upload the
 * image once and decode on the device with hg_hv_unpack_batch_dev (what `hyper-gen dist` / `search` do). */
hg_status hg_sketch_file_read_image(const char *path, hg_sketch_file **out);
const uint8_t *hg_sketch_file_image(const hg_sketch_file *f, size_t *bytes);
uint64_t hg_sketch_file_payload_offset(const hg_sketch_file *f, size_t i);
size_t hg_sketch_file_count(const hg_sketch_file *f);
const hg_file_sketch *hg_sketch_file_get(const hg_sketch_file *f, size_t i);
void hg_sketch_file_free(hg_sketch_file *f);

/* ---- continuous host-fed sketching ------------------------------------------------------------------------
 * The streaming form of hg_sketch_batch for hosts that produce genomes one at a time (reader threads walking a
 * file list, src/sketch_cuda.rs:120-166): one uploader and one compute thread per device keep the PCIe link and the
 * kernels busy without the caller collecting batches.  Results are bit-identical to hg_sketch_batch.
 *   open    one engine per entry of device_ids (an id may repeat); params are fixed for the stream's lifetime
 *   push    any thread; `seq` must stay valid and unchanged until the genome's result has been popped; page-locked
 *           memory (hg_read_fastx_pinned) is fetched by DMA at the link rate, other memory is staged by the runtime;
 *           BLOCKS while hg_sketch_stream_max_pending (4096) results are outstanding -- back-pressure for reader
 *           threads; a caller that pushes and pops on ONE thread must use hg_sketch_stream_try_push instead, which
 *           returns HG_ERR_CAPACITY ("would block": pop some results first) and never waits
 *   pop     any thread; blocks until a result is ready (completion order, identified by `tag`); *got = 0 with
 *           HG_OK once hg_sketch_stream_finish was called and every pushed genome has been popped; hv_out holds
 *           hv_d int16, any output pointer may be NULL
 *   finish  no more pushes; partial chunks are flushed
 *   close   joins the threads and frees everything (outstanding results are dropped)
 * After a failure every call returns its status; hg_sketch_stream_last_error has the text. */
typedef struct hg_sketch_stream hg_sketch_stream;
hg_status hg_sketch_stream_open(const int *device_ids, int n_devices, const hg_sketch_params *p, hg_sketch_stream **out);
hg_status hg_sketch_stream_push(hg_sketch_stream *s, const uint8_t *seq, size_t n_bps, uint64_t tag);
/* the same genome as a hg_pack2 blob (3 bits per base over the link instead of 8; the chunk's blobs go to the
 * packed-input kernels as they arrive, hg_sketch_batch_dev_packed: results are bit-identical).  The blob must have been
 * packed with the stream's norm_mode. */
hg_status hg_sketch_stream_push_packed(hg_sketch_stream *s, const uint8_t *blob, size_t n_bps, uint64_t tag);
/* ... or as a hg_pack2s blob: the codes plus a TABLE of the not-a-base runs instead of the bitmap -- 0.25 bytes per base
 * over the link for an ordinary assembly; the device rebuilds the bitmap.  Results are bit-identical.  blob_bytes = the
 * size of the caller's buffer: the run count is read and hg_pack2s_size(n_bps, n_runs) bytes are uploaded only if they lie
 * inside it, and a table whose runs are not ascending, disjoint, non-empty and inside the sequence is refused
 * (HG_ERR_INVALID) before anything reaches the device. */
hg_status hg_sketch_stream_push_packed_sparse(hg_sketch_stream *s, const uint8_t *blob, size_t blob_bytes, size_t n_bps, uint64_t tag);
/* non-blocking push of any form (packed: 0 = ASCII, 1 = `data` is a hg_pack2 blob, 2 = a hg_pack2s blob -- its run table
 * is validated like above, the buffer's size is the caller's word); HG_ERR_CAPACITY = would block */
hg_status hg_sketch_stream_try_push(hg_sketch_stream *s, const uint8_t *data, size_t n_bps, uint64_t tag, int packed);
size_t hg_sketch_stream_max_pending(const hg_sketch_stream *s);
hg_status hg_sketch_stream_pop(hg_sketch_stream *s, uint64_t *tag, int16_t *hv_out, int32_t *norm2_out,
                               uint32_t *nhash_out, int *got);
hg_status hg_sketch_stream_finish(hg_sketch_stream *s);
const char *hg_sketch_stream_last_error(hg_sketch_stream *s);
/* diagnostics of one engine, seconds since open: out[0] uploader waiting for input, [1] uploader waiting for a free
 * chunk (the kernels lag), [2] uploader inside the copy calls, [3] compute thread waiting for a chunk, [4] compute
 * thread from "chunk taken" to "results queued" (includes waiting for the chunk's upload); out[5] = chunks so far */
hg_status hg_sketch_stream_stats(hg_sketch_stream *s, int engine, double out[6]);
void hg_sketch_stream_close(hg_sketch_stream *s);

/* ---- 2-bit packing for the PCIe link (optional) ----------------------------------------------------
 * hg_pack2 writes hg_pack2_size(n_bps) bytes: 2-bit codes (A,C,G,T = 0..3, 4 bases per byte, LSB first) padded to
 * 16 bytes, then a not-a-base bit per position padded to 16 bytes.  Bases are what the kernels accept under
 * `norm_mode` (ACGTacgt, plus Uu -> T under HG_NORM_U2T).  `out` may equal `seq` when the buffer holds
 * max(n_bps, hg_pack2_size(n_bps)) bytes.  hg_unpack2_dev is the device inverse ('A','C','G','T' / 'N'; 16-byte
 * aligned device pointers, d_seq_out with room for n_bps rounded up to 16), in stream order on the ctx's stream. */
size_t hg_pack2_size(size_t n_bps);
hg_status hg_pack2(const uint8_t *seq, size_t n_bps, uint32_t norm_mode, uint8_t *out);
hg_status hg_unpack2_dev(hg_ctx *ctx, const uint8_t *d_blob, size_t n_bps, uint8_t *d_seq_out);
/* The sparse form for the link: [codes as in hg_pack2, padded to 16 bytes][uint32 n_runs, uint32 0, n_runs x {uint32 first
 * position, uint32 length} of the maximal runs of not-a-base positions, ascending, padded to 16 bytes].  hg_pack2s
 * writes hg_pack2s_size(n_bps, n_runs) bytes to `out` (*size_out) if they fit `cap`; HG_ERR_CAPACITY (*size_out = needed)
 * if not -- a sequence littered with non-bases is better served by hg_pack2 --, HG_ERR_UNSUPPORTED for n_bps >= 2^32.
 * `out` may equal `seq` (cap >= the blob). */
size_t hg_pack2s_size(size_t n_bps, size_t n_runs);
hg_status hg_pack2s(const uint8_t *seq, size_t n_bps, uint32_t norm_mode, uint8_t *out, size_t cap, size_t *size_out);
/* hg_pack2 on the device, byte for byte the host's output: ASCII genomes already in HBM (d_seq + offsets[i], multiples
 * of 4, lens[i] bases) become blobs at d_blobs + blob_offsets[i] (multiples of 16, hg_pack2_size(lens[i]) bytes each) --
 * the input form of hg_sketch_batch_dev_packed.  Host offset arrays, device data; complete on return.
 * hg_pack2_dev: one genome (d_seq 4-byte, d_blob 16-byte aligned). */
hg_status hg_pack2_batch_dev(hg_ctx *ctx, const uint8_t *d_seq, const uint64_t *offsets, const uint64_t *lens, size_t n,
                             uint32_t norm_mode, uint8_t *d_blobs, const uint64_t *blob_offsets);
hg_status hg_pack2_dev(hg_ctx *ctx, const uint8_t *d_seq, size_t n_bps, uint32_t norm_mode, uint8_t *d_blob);

/* ---- FASTA ingest (host side; src/fastx_reader.rs:6-29) ------------------------------- */
/* read_merge_seq: returns a malloc'ed buffer (free with hg_free) and its length */
hg_status hg_read_merge_seq(const char *path, uint8_t **out, size_t *n_bps);
/* same, into a caller-owned growable buffer: *buf (NULL or from malloc/realloc) of *cap bytes is reused when
 * large enough and realloc'ed otherwise -- a reader thread that recycles its buffers avoids one 5 MB
 * mmap / page-fault / munmap cycle per file */
hg_status hg_read_merge_seq_into(const char *path, uint8_t **buf, size_t *cap, size_t *n_bps);
/* mode HG_READ_MERGE: exactly read_merge_seq (every line that does not start with '>' is sequence).
 * mode HG_READ_NEEDLETAIL: what the reference's CPU path sees through needletail 0.5.1 (src/sketch.rs:76-87):
 * FASTA or FASTQ chosen by the first byte ('>' / '@'), four-line FASTQ records (only the sequence line is kept),
 * blanks, tabs and CRs inside sequence lines dropped (`normalize`); still one 'N' per record start.  Both modes
 * inflate gzip input transparently. */
#define HG_READ_MERGE 0u
#define HG_READ_NEEDLETAIL 1u
/* OR-ed into the mode: the buffer comes back as the hg_pack2 blob of the merged sequence (*n_bps = its bases),
 * packed for HG_NORM_ACGT, or for HG_NORM_U2T with HG_READ_PACK2_U2T as well -- for hg_sketch_stream_push_packed */
#define HG_READ_PACK2 16u
#define HG_READ_PACK2_U2T 32u
hg_status hg_read_fastx_into(const char *path, uint32_t mode, uint8_t **buf, size_t *cap, size_t *n_bps);
/* same, but the buffer is page-locked host memory owned by the library (hipHostMalloc, visible to every device;
 * *buf NULL or from an earlier call, release with hg_pinned_free): hg_sketch_batch uploads such sequences by DMA at
 * the PCIe rate, while malloc'ed memory goes through the runtime's bounce buffer at a fraction of it.  Needs a HIP
 * device like every compute entry point. */
hg_status hg_read_fastx_pinned(const char *path, uint32_t mode, uint8_t **buf, size_t *cap, size_t *n_bps);
void hg_pinned_free(void *p);
/* The order in which a dist / Hamming launch with tiles_m x tiles_n tiles of tile_rows x tile_cols walks its tiles (host
 * code only, no device needed; for inspection and for the tests): workgroup slot b runs tile out[b] = tm | tn << 16, or
 * no tile where out[b] == 0xFFFFFFFF.  The hardware deals slot b to XCD b % 8, so the slots b = x, x + 8, ... are XCD
 * x's queue: every tile that has work (with `symmetric`: not entirely on or below the diagonal of the global
 * enumeration, row 0 = ref_off, column 0 = qry_off) exactly once; the XCDs' tile counts differ by at most one; each
 * queue walks 8 x 8 super-tiles in halves of 4 x 8 tiles (the tiles resident on an XCD share 4 + 8 operand blocks through
 * its L2); with diagonal_first the tiles that straddle the diagonal (a database compared with itself has its hits
 * there) lead the queues.  *n_slots = the launch's grid size; HG_ERR_CAPACITY (with *n_slots set) if cap is too small. */
hg_status hg_dist_tile_order(uint32_t tiles_m, uint32_t tiles_n, uint32_t tile_rows, uint32_t tile_cols, int diagonal_first,
                             int symmetric, uint64_t ref_off, uint64_t qry_off, uint32_t *out, size_t cap, size_t *n_slots);
/* NUMA node of a device (-1 unknown): host threads that fill page-locked buffers for it should run there */
int hg_device_numa_node(int device_id);
/* Restricts the CALLING thread to the CPUs of a NUMA node (the ones it may already run on), unless the node has fewer of
 * them than `threads_sharing` (never oversubscribe a node).  Returns 1 when the thread was bound, 0 otherwise (unknown
 * node, nothing readable under /sys, too few CPUs).  Page-locked memory from the HIP runtime lies on the DEVICE's node
 * wherever the allocating thread ran: host threads that read or write such buffers -- the readers of the CLI, the
 * threads of a one-call-per-genome pool, which 2-bit pack into the context's staging buffer -- run ~1.5x faster there
 * than on the other socket (measured: 76-86 against 52-58 GB/s of packing reads, 16 threads). */
int hg_bind_thread_to_numa_node(int node, unsigned threads_sharing);
void hg_free(void *p);

/* ---- bit-packed hypervectors + Hamming search (extension: BASELINE.json configs[4]) ------------
 * No reference counterpart exists (the reference's `search` is an empty stub, src/main.rs:22-24);
 * the semantics are defined here (and restated on the CPU by the test oracle): bit d = (hv[d] >= 0), uint32 word w holds
 * dims 32w..32w+31 LSB first ((hv_d+31)/32 words per vector), distance = popcount(xor).
 * All pointers are device pointers; hv_d must be a multiple of 128 for the search. */
typedef struct {
  uint32_t ref_idx;
  uint32_t qry_idx;
  uint32_t dist;
} hg_ham_hit;
hg_status hg_hv_binarize_dev(hg_ctx *ctx, const int16_t *d_hv, size_t n, uint32_t hv_d, uint32_t *d_bits);
hg_status hg_hamming_full_dev(hg_ctx *ctx, const uint32_t *d_ref_bits, size_t R, const uint32_t *d_qry_bits,
                              size_t Q, uint32_t hv_d, uint32_t *d_dist_out);
/* all pairs with distance <= max_dist, in no particular order; *n_out = found (HG_ERR_CAPACITY if > cap) */
hg_status hg_hamming_search_dev(hg_ctx *ctx, const uint32_t *d_ref_bits, size_t R, const uint32_t *d_qry_bits,
                                size_t Q, uint32_t hv_d, uint32_t max_dist, hg_ham_hit *d_out, size_t cap,
                                size_t *n_out);

/* Searches of 2^24 pairs or more, and every hv_d that is not a multiple of 128, run as an exact GEMM on the matrix pipe
 * with the bits expanded to +-1.0 e2m1 nibbles (G = D - 2 * distance on v_mfma_scale_f32_16x16x128_f8f6f4 with unit
 * block scales; the ANI kernel's tiles and hit lists), smaller ones on the xor + popcount kernel; the +-1 BYTE GEMM
 * (v_mfma_i32_16x16x64_i8) stays reachable through the "ham_path" = "mfma" hook.  All three give the same integers.
 * hg_ctx_last_hamming_path: 0 = xor + popcount, 1 = byte GEMM, 2 = e2m1 (FP4) GEMM (the default for large searches);
 * -1 = none yet. */
int hg_ctx_last_hamming_path(const hg_ctx *ctx);
/* one shard of a sharded reference database: hits carry ref_off + local row, qry_off + local column */
hg_status hg_hamming_search_block_dev(hg_ctx *ctx, const uint32_t *d_ref_bits, size_t R, size_t ref_off,
                                      const uint32_t *d_qry_bits, size_t Q, size_t qry_off, uint32_t hv_d,
                                      uint32_t max_dist, hg_ham_hit *d_out, size_t cap, size_t *n_out);

/* ---- several GPUs in one process -------------------------------------------------------------------
 * The reference drives one CudaDevice::new(0) from all rayon workers (src/sketch_cuda.rs:52,82).  hg_multi
 * owns one hg_ctx per entry of device_ids (1..8 GPUs of one node; ids may repeat -- two logical shards on one
 * GPU, which is how a single-GPU box tests the sharded path) and one host thread per shard for the duration
 * of each call.  Partitioning follows SURVEY.md 8(e):
 *   sketch : genomes are independent units -> contiguous genome blocks per shard, no exchange;
 *   dist   : every shard needs ALL reference HVs -> the shards' row blocks are all-gathered device to
 *            device (hipMemcpyPeerAsync: every GPU pulls its 7 peers' blocks over its 7 xGMI links at
 *            once), then shard s computes the block (all refs) x (its query rows); hits are merged on
 *            the host with global indices;
 *   search : the bit-packed reference database is sharded by rows, the (small) query set is broadcast,
 *            per-shard hits are merged on the host.
 * Results equal the single-ctx calls (same kernels, same arithmetic); hit order is unspecified. */
typedef struct hg_multi hg_multi;
hg_status hg_multi_create(const int *device_ids, int n, hg_multi **out);
void hg_multi_destroy(hg_multi *m);
int hg_multi_size(const hg_multi *m);
hg_ctx *hg_multi_ctx(hg_multi *m, int shard); /* borrowed: valid until hg_multi_destroy */
const char *hg_multi_last_error(const hg_multi *m);
/* what hipDeviceEnablePeerAccess said when the shards were opened ("peer access enabled for 56 of 56 ordered device
 * pairs", or the pairs that fall back to host-staged copies and why) */
const char *hg_multi_peer_report(const hg_multi *m);
/* The exchange step of dist (the reference has none: src/dist.rs:231-294 runs on one host).
 *   HG_GATHER_PEER (default): direct hipMemcpyPeerAsync pulls, one per peer block;
 *   HG_GATHER_RCCL          : ncclAllGather (equal row blocks) / grouped ncclBroadcast per owner over one RCCL
 *                             communicator per shard (ncclCommInitAll).  Needs distinct device ids; librccl is opened
 *                             with dlopen at this call (HG_ERR_UNSUPPORTED if it is absent).
 * Both give the same gathered matrix, hence the same hits.  hg_multi_gather_report: how the last exchange ran. */
enum { HG_GATHER_PEER = 0, HG_GATHER_RCCL = 1 };
hg_status hg_multi_set_gather(hg_multi *m, int mode);
int hg_multi_gather_mode(const hg_multi *m);
const char *hg_multi_gather_report(const hg_multi *m);
/* contiguous block [*lo, *hi) of n units owned by `shard` of n_shards; sizes differ by at most one */
void hg_shard_range(size_t n, int shard, int n_shards, size_t *lo, size_t *hi);

/* hg_sketch_batch over all shards: genome g goes to the shard whose hg_shard_range contains g */
hg_status hg_sketch_batch_multi(hg_multi *m, const uint8_t *const *seqs, const size_t *lens, size_t n,
                                const hg_sketch_params *p, int16_t *hv_out, int32_t *norm2_out,
                                uint32_t *nhash_out);
/* hg_dist over all shards, host buffers in and out.  Each reference row crosses PCIe once (to its shard's
 * GPU) and is replicated over xGMI; query rows go to their shard only.  qry_hv == ref_hv with Q == R is the
 * all-vs-all case: every GPU then holds every row, and with `symmetric` the column ranges are balanced by
 * pair count (boundaries at Q * sqrt(s / n)) instead of by row count. */
hg_status hg_dist_multi(hg_multi *m, const int16_t *ref_hv, const int32_t *ref_norm2, size_t R,
                        const int16_t *qry_hv, const int32_t *qry_norm2, size_t Q, uint32_t hv_d,
                        uint32_t ksize, int symmetric, float ani_th, hg_ani_hit *out, size_t cap,
                        size_t *n_out);
/* the same for sketches that are already resident: shard s holds ref_rows[s] consecutive reference rows at
 * d_ref_hv[s] / d_ref_norm2[s] on ITS device (e.g. the output of hg_sketch_batch_dev on hg_multi_ctx(m, s)),
 * global row order = shard order.  d_qry_hv == NULL: the query set is the reference set (all-vs-all). */
hg_status hg_dist_multi_dev(hg_multi *m, const int16_t *const *d_ref_hv, const int32_t *const *d_ref_norm2,
                            const size_t *ref_rows, const int16_t *const *d_qry_hv,
                            const int32_t *const *d_qry_norm2, const size_t *qry_rows, uint32_t hv_d,
                            uint32_t ksize, int symmetric, float ani_th, hg_ani_hit *out, size_t cap,
                            size_t *n_out);
/* bit-packed database search (host buffers): ref_bits is R x words, qry_bits Q x words, words = hv_d / 32 */
hg_status hg_hamming_search_multi(hg_multi *m, const uint32_t *ref_bits, size_t R, const uint32_t *qry_bits,
                                  size_t Q, uint32_t hv_d, uint32_t max_dist, hg_ham_hit *out, size_t cap,
                                  size_t *n_out);

/* ---- synthetic genomes (benchmark / test utility, not part of the reference surface) -----
 * Genome g = first_genome + i is written at d_out + i * stride as 'N' followed by L bases
 * (the read_merge_seq layout of a one-record FASTA).  Cluster c = g / cluster_size is an iid
 * uniform ACGT root; member m = g % cluster_size has iid substitutions at rate
 * m * sub_ppm_per_member / 1e6.  stride >= L + 1; n <= 65535 per call. */
hg_status hg_synth_genomes_dev(hg_ctx *ctx, uint64_t first_genome, size_t n, uint64_t L,
                               uint32_t cluster_size, uint32_t sub_ppm_per_member, uint64_t stride,
                               uint8_t *d_out);

#ifdef __cplusplus
}
#endif
#endif /* HYPERGEN_H */
