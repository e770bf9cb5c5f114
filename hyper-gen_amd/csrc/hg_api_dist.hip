// hg_api_dist.hip -- the C ABI of include/hypergen.h, part 3: the dist path -- full and thresholded ANI matrices of device- or
// host-resident hypervectors, blocks of a larger matrix, operands prepared where the rows live (src/dist.rs:139-161,231-294).
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
// dist
// ---------------------------------------------------------------------------------------------
namespace {
hg_status check_dist(hg_ctx *c, size_t R, size_t Q, uint32_t hv_d, uint32_t ksize) {
  if (R > 0x7FFFFFFFull || Q > 0x7FFFFFFFull) return hg_fail(c, HG_ERR_UNSUPPORTED, "R, Q must be < 2^31");
  if (hv_d == 0 || hv_d > 65536) return hg_fail(c, HG_ERR_UNSUPPORTED, "hv_d must be in 1..65536");
  if (ksize == 0) return hg_fail(c, HG_ERR_INVALID, "ksize must be >= 1");
  return HG_OK;
}
}  // namespace

extern "C" hg_status hg_dist_full_dev(hg_ctx *c, const int16_t *d_ref_hv, const int32_t *d_ref_norm2, size_t R,
                                      const int16_t *d_qry_hv, const int32_t *d_qry_norm2, size_t Q, uint32_t hv_d,
                                      uint32_t ksize, float *d_ani_out) {
  if (!c) return HG_ERR_INVALID;
  hg_status s = check_dist(c, R, Q, hv_d, ksize);
  if (s != HG_OK) return s;
  if (R == 0 || Q == 0) return HG_OK;
  if (!d_ref_hv || !d_ref_norm2 || !d_qry_hv || !d_qry_norm2 || !d_ani_out) return hg_fail(c, HG_ERR_INVALID, "NULL argument");
  HG_ENTER(c);
  hg_dist_args a{};
  a.ref_hv = d_ref_hv, a.ref_n2 = d_ref_norm2, a.qry_hv = d_qry_hv, a.qry_n2 = d_qry_norm2;
  a.R = (uint32_t)R, a.Q = (uint32_t)Q, a.hv_d = hv_d, a.ksize = ksize;
  a.ani_out = d_ani_out;
  return hg_run_dist(c, a);
}

extern "C" hg_status hg_dist_dev(hg_ctx *c, const int16_t *d_ref_hv, const int32_t *d_ref_norm2, size_t R,
                                 const int16_t *d_qry_hv, const int32_t *d_qry_norm2, size_t Q, uint32_t hv_d,
                                 uint32_t ksize, int symmetric, float ani_th, hg_ani_hit *d_out, size_t cap,
                                 size_t *n_out) {
  return hg_dist_block_dev(c, d_ref_hv, d_ref_norm2, R, 0, d_qry_hv, d_qry_norm2, Q, 0, hv_d, ksize, symmetric, ani_th,
                           d_out, cap, n_out);
}

static hg_status dist_block_once(hg_ctx *c, const int16_t *d_ref_hv, const int32_t *d_ref_norm2, size_t R, size_t ref_off,
                                 const int16_t *d_qry_hv, const int32_t *d_qry_norm2, size_t Q, size_t qry_off, uint32_t hv_d,
                                 uint32_t ksize, int symmetric, float ani_th, hg_ani_hit *d_out, size_t cap, size_t *n_out);

extern "C" hg_status hg_dist_block_dev(hg_ctx *c, const int16_t *d_ref_hv, const int32_t *d_ref_norm2, size_t R,
                                       size_t ref_off, const int16_t *d_qry_hv, const int32_t *d_qry_norm2, size_t Q,
                                       size_t qry_off, uint32_t hv_d, uint32_t ksize, int symmetric, float ani_th,
                                       hg_ani_hit *d_out, size_t cap, size_t *n_out) {
  // The kernels count hits in 32 bits.  A comparison of more than 2^32 - 1 pairs (66 000 x 66 000 and up) could report more
  // hits than that with a low threshold, so it runs as blocks of reference rows with fewer pairs each -- global indices, the
  // i < j rule and the hit list are those of the one call; the counts add up in 64 bits, and once the caller's buffer is full
  // the remaining blocks only count (the contract: *n_out = all hits found, HG_ERR_CAPACITY if they did not fit).
  const uint64_t pair_limit = (c && c->dbg_pair_limit) ? c->dbg_pair_limit : 0xFFFFFFFFull;  // ("pair_limit": test hook)
  if (c && n_out && Q && (uint64_t)R * (uint64_t)Q > pair_limit && R <= 0x7FFFFFFFull && Q <= 0x7FFFFFFFull) {
    const size_t rows_per = std::max<size_t>(1, (size_t)(pair_limit / (uint64_t)Q));
    size_t total = 0;
    bool full = false;
    *n_out = 0;
    for (size_t r0 = 0; r0 < R; r0 += rows_per) {
      const size_t rows = std::min(rows_per, R - r0), room = total < cap ? cap - total : 0;
      size_t got = 0;
      const hg_status bs = dist_block_once(c, d_ref_hv + r0 * (size_t)hv_d, d_ref_norm2 + r0, rows, ref_off + r0, d_qry_hv, d_qry_norm2, Q,
                                           qry_off, hv_d, ksize, symmetric, ani_th, d_out ? d_out + std::min(total, cap) : nullptr, room, &got);
      if (bs == HG_ERR_CAPACITY) full = true;
      else if (bs != HG_OK) return bs;
      total += got;
    }
    *n_out = total;
    if (full || total > cap) return hg_fail(c, HG_ERR_CAPACITY, "hit buffer too small");
    return HG_OK;
  }
  return dist_block_once(c, d_ref_hv, d_ref_norm2, R, ref_off, d_qry_hv, d_qry_norm2, Q, qry_off, hv_d, ksize, symmetric, ani_th, d_out,
                         cap, n_out);
}

static hg_status dist_block_once(hg_ctx *c, const int16_t *d_ref_hv, const int32_t *d_ref_norm2, size_t R, size_t ref_off,
                                 const int16_t *d_qry_hv, const int32_t *d_qry_norm2, size_t Q, size_t qry_off, uint32_t hv_d,
                                 uint32_t ksize, int symmetric, float ani_th, hg_ani_hit *d_out, size_t cap, size_t *n_out) {
  if (!c) return HG_ERR_INVALID;
  if (!n_out) return hg_fail(c, HG_ERR_INVALID, "n_out == NULL");
  *n_out = 0;
  hg_status s = check_dist(c, R, Q, hv_d, ksize);
  if (s != HG_OK) return s;
  if (ref_off + R > 0x7FFFFFFFull || qry_off + Q > 0x7FFFFFFFull) return hg_fail(c, HG_ERR_UNSUPPORTED, "global indices must be < 2^31");
  if (R == 0 || Q == 0) return HG_OK;
  if (!d_ref_hv || !d_ref_norm2 || !d_qry_hv || !d_qry_norm2 || (cap && !d_out)) return hg_fail(c, HG_ERR_INVALID, "NULL argument");
  HG_ENTER(c);
  if ((s = hg_ensure(c, c->w_misc, 64)) != HG_OK) return s;
  // [0] hit counter, [1] exactness verdict, [2] its window length, [4..12] control words of the i8 operand attempt
  auto *d_count = static_cast<uint32_t *>(c->w_misc.p);
  // (zeroed by the previous call on its way out, behind its read-back: one command less in front of the kernels;
  // the first call, and one after a call that failed half way, does it here)
  if (c->misc_zeroed != d_count) HG_HIP(c, hipMemsetAsync(d_count, 0, 16 * sizeof(uint32_t), c->stream));
  c->misc_zeroed = nullptr;
  hg_dist_args a{};
  a.ref_hv = d_ref_hv, a.ref_n2 = d_ref_norm2, a.qry_hv = d_qry_hv, a.qry_n2 = d_qry_norm2;
  a.R = (uint32_t)R, a.Q = (uint32_t)Q, a.hv_d = hv_d, a.ksize = ksize;
  a.hits = d_out, a.hit_count = d_count;
  a.hit_cap = cap > 0xFFFFFFFFull ? 0xFFFFFFFFu : (uint32_t)cap;
  a.ani_th = ani_th, a.symmetric = symmetric;
  a.ref_off = (uint32_t)ref_off, a.qry_off = (uint32_t)qry_off;
  int spec_cover = -1;
  if ((s = hg_run_dist(c, a, d_count + 1, &spec_cover)) != HG_OK) return s;
  const uint32_t *h_res = nullptr;
  // (the publishing kernel clears the control words behind the copy: they are ready for a second pass, or for the next call)
  if ((s = hg_publish_words(c, d_count, 16, &h_res, 16)) != HG_OK) return s;
  c->misc_zeroed = d_count;
  const bool i8_tried = h_res[9] != 0;  // the i8 prepass wrote its K-step count
  if (i8_tried && h_res[8] != 1u) c->i8_skip = 16;  // vetoed on the device: f16 ran; do not probe again for a while
  if (h_res[8] == 1u) {
    c->last_dist_path = 1;
    c->last_kernel[HG_T_DIST] = c->last_kernel_i8;
    c->i8_sig_ref = d_ref_hv, c->i8_sig_qry = d_qry_hv, c->i8_sig_r = (uint32_t)R, c->i8_sig_q = (uint32_t)Q, c->i8_sig_d = hv_d;
  } else {
    c->i8_sig_ref = c->i8_sig_qry = nullptr;
  }
  if (h_res[8] == 2u) {  // the centred f16 kernel did the work
    c->last_dist_path = 3;
    c->last_kernel[HG_T_DIST] = c->last_kernel_cen;
    c->cen_sig_ref = d_ref_hv, c->cen_sig_qry = d_qry_hv, c->cen_sig_r = (uint32_t)R, c->cen_sig_q = (uint32_t)Q, c->cen_sig_d = hv_d;
  } else {
    c->cen_sig_ref = c->cen_sig_qry = nullptr;
  }
  // no guarded launch applied (or the raw f16 chain was not queued behind a trusted i8 / centred attempt that failed after
  // all): statistics-driven schedule
  if (h_res[8] == 0u && (spec_cover == -2 || (spec_cover >= 0 && (int)h_res[1] > spec_cover))) {
    c->misc_zeroed = nullptr;
    if ((s = hg_run_dist(c, a)) != HG_OK) return s;
    if ((s = hg_publish_words(c, d_count, 16, &h_res, 16)) != HG_OK) return s;
    c->misc_zeroed = d_count;
  }
  const uint32_t found = h_res[0];
  *n_out = found;
  if (found > cap) return hg_fail(c, HG_ERR_CAPACITY, "hit buffer too small");
  return HG_OK;
}

// ---- sharded dist: the reference operands are prepared where the rows live (the reference has no such step) ----
extern "C" size_t hg_dist_ops_row_bytes(uint32_t hv_d) { return hg_dist_ops_row_bytes_impl(hv_d); }
extern "C" size_t hg_dist_ops_meta_bytes(void) { return hg_dist_ops_meta_bytes_impl(); }
extern "C" size_t hg_dist_ops_padded_rows(size_t rows) { return hg_dist_ops_padded_rows_impl(rows); }

extern "C" hg_status hg_dist_prep_ops_dev(hg_ctx *c, const int16_t *d_hv, size_t rows, uint32_t hv_d, uint8_t *d_ops,
                                          uint8_t *d_meta, uint32_t *d_flag) {
  if (!c) return HG_ERR_INVALID;
  if (rows == 0) return HG_OK;
  if (!d_hv || !d_ops || !d_meta || !d_flag || hv_d == 0) return hg_fail(c, HG_ERR_INVALID, "NULL argument");
  if (rows > 0x7FFFFFFFull) return hg_fail(c, HG_ERR_UNSUPPORTED, "more than 2^31 rows");
  HG_ENTER(c);
  return hg_run_dist_prep_ops(c, d_hv, (uint32_t)rows, hv_d, d_ops, d_meta, d_flag);
}

static hg_status dist_block_ops_once(hg_ctx *c, const uint8_t *d_ref_ops, const uint8_t *d_ref_meta, const int32_t *d_ref_norm2, size_t R,
                                     size_t ref_off, const uint32_t *d_ref_index, const uint32_t *d_flags, size_t n_flags,
                                     const int16_t *d_qry_hv, const int32_t *d_qry_norm2, size_t Q, size_t qry_off, uint32_t hv_d,
                                     uint32_t ksize, int symmetric, float ani_th, hg_ani_hit *d_out, size_t cap, size_t *n_out);

extern "C" hg_status hg_dist_block_ops_dev(hg_ctx *c, const uint8_t *d_ref_ops, const uint8_t *d_ref_meta, const int32_t *d_ref_norm2,
                                           size_t R, size_t ref_off, const uint32_t *d_ref_index, const uint32_t *d_flags,
                                           size_t n_flags, const int16_t *d_qry_hv, const int32_t *d_qry_norm2, size_t Q,
                                           size_t qry_off, uint32_t hv_d, uint32_t ksize, int symmetric, float ani_th,
                                           hg_ani_hit *d_out, size_t cap, size_t *n_out) {
  // (32-bit hit counter, see hg_dist_block_dev; here the prepared reference block stays whole -- its padding rows belong to
  // it -- and the QUERY rows go in blocks)
  const uint64_t pair_limit = (c && c->dbg_pair_limit) ? c->dbg_pair_limit : 0xFFFFFFFFull;
  if (c && n_out && R && (uint64_t)R * (uint64_t)Q > pair_limit && R <= 0x7FFFFFFFull && Q <= 0x7FFFFFFFull) {
    const size_t cols_per = std::max<size_t>(1, (size_t)(pair_limit / (uint64_t)R));
    size_t total = 0;
    bool full = false;
    *n_out = 0;
    for (size_t q0 = 0; q0 < Q; q0 += cols_per) {
      const size_t cols = std::min(cols_per, Q - q0), room = total < cap ? cap - total : 0;
      size_t got = 0;
      const hg_status bs = dist_block_ops_once(c, d_ref_ops, d_ref_meta, d_ref_norm2, R, ref_off, d_ref_index, d_flags, n_flags,
                                               d_qry_hv + q0 * (size_t)hv_d, d_qry_norm2 + q0, cols, qry_off + q0, hv_d, ksize, symmetric, ani_th,
                                               d_out ? d_out + std::min(total, cap) : nullptr, room, &got);
      if (bs == HG_ERR_CAPACITY) full = true;
      else if (bs != HG_OK) return bs;  // (HG_ERR_INEXACT included: the caller falls back for the whole call)
      total += got;
    }
    *n_out = total;
    if (full || total > cap) return hg_fail(c, HG_ERR_CAPACITY, "hit buffer too small");
    return HG_OK;
  }
  return dist_block_ops_once(c, d_ref_ops, d_ref_meta, d_ref_norm2, R, ref_off, d_ref_index, d_flags, n_flags, d_qry_hv, d_qry_norm2, Q,
                             qry_off, hv_d, ksize, symmetric, ani_th, d_out, cap, n_out);
}

static hg_status dist_block_ops_once(hg_ctx *c, const uint8_t *d_ref_ops, const uint8_t *d_ref_meta, const int32_t *d_ref_norm2, size_t R,
                                     size_t ref_off, const uint32_t *d_ref_index, const uint32_t *d_flags, size_t n_flags,
                                     const int16_t *d_qry_hv, const int32_t *d_qry_norm2, size_t Q, size_t qry_off, uint32_t hv_d,
                                     uint32_t ksize, int symmetric, float ani_th, hg_ani_hit *d_out, size_t cap, size_t *n_out) {
  if (!c) return HG_ERR_INVALID;
  if (!n_out) return hg_fail(c, HG_ERR_INVALID, "n_out == NULL");
  *n_out = 0;
  hg_status s = check_dist(c, R, Q, hv_d, ksize);
  if (s != HG_OK) return s;
  if (ref_off + R > 0x7FFFFFFFull || qry_off + Q > 0x7FFFFFFFull) return hg_fail(c, HG_ERR_UNSUPPORTED, "global indices must be < 2^31");
  if (R == 0 || Q == 0) return HG_OK;
  if (!d_ref_ops || !d_ref_meta || !d_ref_norm2 || !d_qry_hv || !d_qry_norm2 || (cap && !d_out) || (n_flags && !d_flags))
    return hg_fail(c, HG_ERR_INVALID, "NULL argument");
  if (d_ref_index && symmetric) return hg_fail(c, HG_ERR_UNSUPPORTED, "symmetric needs contiguous reference indices (no d_ref_index)");
  HG_ENTER(c);
  if ((s = hg_ensure(c, c->w_misc, 64)) != HG_OK) return s;
  auto *d_count = static_cast<uint32_t *>(c->w_misc.p);
  if (c->misc_zeroed != d_count) HG_HIP(c, hipMemsetAsync(d_count, 0, 16 * sizeof(uint32_t), c->stream));
  c->misc_zeroed = nullptr;
  hg_dist_args a{};
  a.ref_n2 = d_ref_norm2, a.qry_hv = d_qry_hv, a.qry_n2 = d_qry_norm2;
  a.R = (uint32_t)R, a.Q = (uint32_t)Q, a.hv_d = hv_d, a.ksize = ksize;
  a.hits = d_out, a.hit_count = d_count;
  a.hit_cap = cap > 0xFFFFFFFFull ? 0xFFFFFFFFu : (uint32_t)cap;
  a.ani_th = ani_th, a.symmetric = symmetric;
  a.ref_off = (uint32_t)ref_off, a.qry_off = (uint32_t)qry_off;
  a.ref_ops = d_ref_ops, a.ref_meta = d_ref_meta, a.ref_flags = d_flags, a.n_flags = (uint32_t)n_flags, a.ref_index = d_ref_index;
  int spec = -1;
  if ((s = hg_run_dist(c, a, d_count + 1, &spec)) != HG_OK) return s;
  const uint32_t *h_res = nullptr;
  if ((s = hg_publish_words(c, d_count, 16, &h_res, 16)) != HG_OK) return s;
  c->misc_zeroed = d_count;
  const bool valid = h_res[8] == 1u;
  const uint32_t found = h_res[0];
  c->i8_sig_ref = c->i8_sig_qry = nullptr;
  if (!valid) {
    // an owner's rows, or this call's query rows, do not fit the byte-operand scheme (mixed parity, a residual beyond a
    // byte, more clamped entries than a row's slots): nothing was reported; the caller gathers the i16 rows instead
    return hg_fail(c, HG_ERR_INEXACT, "prepared operands vetoed on the device: fall back to hg_dist_block_dev on the i16 rows");
  }
  c->last_dist_path = 1;
  c->last_kernel[HG_T_DIST] = c->last_kernel_i8;
  *n_out = found;
  if (found > cap) return hg_fail(c, HG_ERR_CAPACITY, "hit buffer too small");
  return HG_OK;
}

namespace {
struct StagedDist {
  const int16_t *d_ref, *d_qry;
  const int32_t *d_rn, *d_qn;
};
hg_status stage_dist(hg_ctx *c, const int16_t *ref_hv, const int32_t *ref_n2, size_t R, const int16_t *qry_hv,
                     const int32_t *qry_n2, size_t Q, uint32_t hv_d, StagedDist &o) {
  hg_status s;
  const size_t rb = R * (size_t)hv_d * 2, qb = Q * (size_t)hv_d * 2;
  // a set compared with itself (src/dist.rs:13, path_r == path_q) travels once, and the device path sees one matrix:
  // one operand prepass instead of two, the diagonal tiles of the GEMM first
  const bool same = ref_hv == qry_hv && ref_n2 == qry_n2 && R == Q;
  if ((s = hg_ensure(c, c->w_hv, rb + 64)) != HG_OK) return s;
  if (!same && (s = hg_ensure(c, c->w_hv2, qb + 64)) != HG_OK) return s;
  if ((s = hg_ensure(c, c->w_n2a, R * 4 + 64)) != HG_OK) return s;
  if (!same && (s = hg_ensure(c, c->w_n2b, Q * 4 + 64)) != HG_OK) return s;
  HG_HIP(c, hipMemcpyAsync(c->w_hv.p, ref_hv, rb, hipMemcpyHostToDevice, c->stream));
  if (!same) HG_HIP(c, hipMemcpyAsync(c->w_hv2.p, qry_hv, qb, hipMemcpyHostToDevice, c->stream));
  HG_HIP(c, hipMemcpyAsync(c->w_n2a.p, ref_n2, R * 4, hipMemcpyHostToDevice, c->stream));
  if (!same) HG_HIP(c, hipMemcpyAsync(c->w_n2b.p, qry_n2, Q * 4, hipMemcpyHostToDevice, c->stream));
  o.d_ref = static_cast<int16_t *>(c->w_hv.p), o.d_qry = same ? o.d_ref : static_cast<int16_t *>(c->w_hv2.p);
  o.d_rn = static_cast<int32_t *>(c->w_n2a.p), o.d_qn = same ? o.d_rn : static_cast<int32_t *>(c->w_n2b.p);
  return HG_OK;
}
}  // namespace

extern "C" hg_status hg_dist_full(hg_ctx *c, const int16_t *ref_hv, const int32_t *ref_norm2, size_t R,
                                  const int16_t *qry_hv, const int32_t *qry_norm2, size_t Q, uint32_t hv_d,
                                  uint32_t ksize, float *ani_out) {
  if (!c) return HG_ERR_INVALID;
  hg_status s = check_dist(c, R, Q, hv_d, ksize);
  if (s != HG_OK) return s;
  if (R == 0 || Q == 0) return HG_OK;
  if (!ref_hv || !ref_norm2 || !qry_hv || !qry_norm2 || !ani_out) return hg_fail(c, HG_ERR_INVALID, "NULL argument");
  HG_ENTER(c);
  StagedDist sd;
  if ((s = stage_dist(c, ref_hv, ref_norm2, R, qry_hv, qry_norm2, Q, hv_d, sd)) != HG_OK) return s;
  if ((s = hg_ensure(c, c->w_ani, R * Q * sizeof(float) + 64)) != HG_OK) return s;
  s = hg_dist_full_dev(c, sd.d_ref, sd.d_rn, R, sd.d_qry, sd.d_qn, Q, hv_d, ksize, static_cast<float *>(c->w_ani.p));
  if (s != HG_OK) return s;
  HG_HIP(c, hipMemcpyAsync(ani_out, c->w_ani.p, R * Q * sizeof(float), hipMemcpyDeviceToHost, c->stream));
  HG_HIP(c, hipStreamSynchronize(c->stream));
  return HG_OK;
}

extern "C" hg_status hg_dist(hg_ctx *c, const int16_t *ref_hv, const int32_t *ref_norm2, size_t R,
                             const int16_t *qry_hv, const int32_t *qry_norm2, size_t Q, uint32_t hv_d, uint32_t ksize,
                             int symmetric, float ani_th, hg_ani_hit *out, size_t cap, size_t *n_out) {
  if (!c) return HG_ERR_INVALID;
  if (!n_out) return hg_fail(c, HG_ERR_INVALID, "n_out == NULL");
  *n_out = 0;
  hg_status s = check_dist(c, R, Q, hv_d, ksize);
  if (s != HG_OK) return s;
  if (R == 0 || Q == 0) return HG_OK;
  if (!ref_hv || !ref_norm2 || !qry_hv || !qry_norm2 || (cap && !out)) return hg_fail(c, HG_ERR_INVALID, "NULL argument");
  HG_ENTER(c);
  StagedDist sd;
  if ((s = stage_dist(c, ref_hv, ref_norm2, R, qry_hv, qry_norm2, Q, hv_d, sd)) != HG_OK) return s;
  if ((s = hg_ensure(c, c->w_ani, cap * sizeof(hg_ani_hit) + 64)) != HG_OK) return s;
  size_t found = 0;
  s = hg_dist_dev(c, sd.d_ref, sd.d_rn, R, sd.d_qry, sd.d_qn, Q, hv_d, ksize, symmetric, ani_th,
                  static_cast<hg_ani_hit *>(c->w_ani.p), cap, &found);
  *n_out = found;
  if (s != HG_OK && s != HG_ERR_CAPACITY) return s;
  const size_t ncopy = std::min(found, cap);
  if (ncopy) {
    HG_HIP(c, hipMemcpyAsync(out, c->w_ani.p, ncopy * sizeof(hg_ani_hit), hipMemcpyDeviceToHost, c->stream));
    HG_HIP(c, hipStreamSynchronize(c->stream));
  }
  return s;
}

extern "C" void hg_sort_ani_hits(hg_ani_hit *hits, size_t n, size_t Q, int symmetric) {
  // dump_ani_file (src/utils.rs:262-269): stable ascending sort by ANI over the enumeration
  // order (row-major, src/dist.rs:251-265), then reversed => descending ANI, ties in reverse
  // enumeration order.
  (void)symmetric;
  auto key = [Q](const hg_ani_hit &h) { return (uint64_t)h.ref_idx * Q + h.qry_idx; };
  std::sort(hits, hits + n, [&](const hg_ani_hit &a, const hg_ani_hit &b) {
    if (a.ani != b.ani) return a.ani > b.ani;
    return key(a) > key(b);
  });
}
