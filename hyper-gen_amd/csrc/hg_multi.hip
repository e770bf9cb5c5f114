// hg_multi.hip -- several GPUs in one process behind the C ABI (include/hypergen.h, "several GPUs in one
// process").  Host code only: every shard is an ordinary hg_ctx and runs the single-GPU entry points; this file
// adds the partitioning of SURVEY.md 8(e), the device-to-device all-gather of the reference HV matrix and the
// merge of the per-shard hit lists.
//
// Exchange step.  MI355X boards are fully connected by xGMI (7 links per GPU), so the all-gather is done as
// direct pulls: every GPU copies each peer's row block straight into its own gathered matrix with
// hipMemcpyPeerAsync -- 7 concurrent transfers per GPU, one per link, each block crossing exactly one link.
// (A ring all-gather would push every block over n - 1 hops of a per-link-bound ring.)  Ordering is by events:
// a block is published by an event on its owner's stream, the puller's stream waits for that event.
#include <algorithm>
#include <cmath>
#include <cstring>
#include <thread>
#include <vector>

#include "hg_internal.h"

struct hg_multi {
  std::vector<hg_ctx *> ctx;
  std::vector<int> dev;
  std::string err;
  struct Shard {
    hg_ctx::Buf ref_all, n2_all;  // gathered reference matrix + norms (every shard holds all rows)
    hg_ctx::Buf mine, mine_n2;    // this shard's uploaded reference rows (host entry points)
    hg_ctx::Buf qry, qry_n2;      // this shard's query rows (host entry points)
    hg_ctx::Buf hits;             // per-shard hit list
    hipEvent_t ready = nullptr;   // "this shard's published rows are complete"
  };
  std::vector<Shard> sh;
};

namespace {

hg_status mfail(hg_multi *m, hg_status s, const std::string &msg) {
  if (m) m->err = msg;
  return s;
}

// run fn(shard) on one host thread per shard; the first failing status wins
template <class F>
hg_status for_each_shard(hg_multi *m, F &&fn) {
  const int n = (int)m->ctx.size();
  std::vector<hg_status> st(n, HG_OK);
  if (n == 1) {
    st[0] = fn(0);
  } else {
    std::vector<std::thread> th;
    th.reserve(n);
    for (int s = 0; s < n; ++s) th.emplace_back([&, s] { st[s] = fn(s); });
    for (auto &t : th) t.join();
  }
  for (int s = 0; s < n; ++s)
    if (st[s] != HG_OK && st[s] != HG_ERR_CAPACITY)
      return mfail(m, st[s], "shard " + std::to_string(s) + " (device " + std::to_string(m->dev[s]) + "): " +
                                 hg_last_error(m->ctx[s]));
  for (int s = 0; s < n; ++s)
    if (st[s] == HG_ERR_CAPACITY) return HG_ERR_CAPACITY;
  return HG_OK;
}

// dst (on shard d) <- src (on shard s), ordered on shard d's stream
hipError_t peer_copy(hg_multi *m, int d, void *dst, int s, const void *src, size_t bytes) {
  if (!bytes) return hipSuccess;
  if (m->dev[d] == m->dev[s]) return hipMemcpyAsync(dst, src, bytes, hipMemcpyDeviceToDevice, m->ctx[d]->stream);
  return hipMemcpyPeerAsync(dst, m->dev[d], src, m->dev[s], bytes, m->ctx[d]->stream);
}

}  // namespace

extern "C" void hg_shard_range(size_t n, int shard, int n_shards, size_t *lo, size_t *hi) {
  if (n_shards < 1) n_shards = 1;
  if (shard < 0) shard = 0;
  if (shard >= n_shards) shard = n_shards - 1;
  const size_t base = n / (size_t)n_shards, rem = n % (size_t)n_shards, s = (size_t)shard;
  const size_t l = s * base + std::min(s, rem);
  if (lo) *lo = l;
  if (hi) *hi = l + base + (s < rem ? 1 : 0);
}

extern "C" hg_status hg_multi_create(const int *device_ids, int n, hg_multi **out) {
  if (!out) return HG_ERR_INVALID;
  *out = nullptr;
  if (!device_ids || n < 1 || n > 64) return hg_fail(nullptr, HG_ERR_INVALID, "device_ids: 1..64 entries");
  hg_multi *m = new (std::nothrow) hg_multi();
  if (!m) return hg_fail(nullptr, HG_ERR_OOM, "hg_multi allocation");
  m->sh.resize(n);
  for (int s = 0; s < n; ++s) {
    hg_ctx *c = nullptr;
    const hg_status st = hg_ctx_create(device_ids[s], &c);
    if (st != HG_OK) {
      hg_multi_destroy(m);
      return st;  // message already in the thread's creation-error slot
    }
    m->ctx.push_back(c), m->dev.push_back(device_ids[s]);
    if (hipSetDevice(device_ids[s]) != hipSuccess ||
        hipEventCreateWithFlags(&m->sh[s].ready, hipEventDisableTiming) != hipSuccess) {
      hg_multi_destroy(m);
      return hg_fail(nullptr, HG_ERR_HIP, "hg_multi: event creation failed");
    }
  }
  // direct xGMI access between distinct devices (hipMemcpyPeerAsync stages through the host without it)
  for (int a = 0; a < n; ++a)
    for (int b = 0; b < n; ++b) {
      if (m->dev[a] == m->dev[b]) continue;
      int can = 0;
      if (hipDeviceCanAccessPeer(&can, m->dev[a], m->dev[b]) == hipSuccess && can) {
        (void)hipSetDevice(m->dev[a]);
        const hipError_t e = hipDeviceEnablePeerAccess(m->dev[b], 0);
        if (e != hipSuccess) (void)hipGetLastError();  // already enabled is fine
      }
    }
  *out = m;
  return HG_OK;
}

extern "C" void hg_multi_destroy(hg_multi *m) {
  if (!m) return;
  for (size_t s = 0; s < m->ctx.size(); ++s) {
    (void)hipSetDevice(m->dev[s]);
    (void)hipStreamSynchronize(m->ctx[s]->stream);
    hg_multi::Shard &x = m->sh[s];
    for (hg_ctx::Buf *b : {&x.ref_all, &x.n2_all, &x.mine, &x.mine_n2, &x.qry, &x.qry_n2, &x.hits})
      if (b->p) (void)hipFree(b->p);
  }
  for (size_t s = 0; s < m->sh.size(); ++s)
    if (m->sh[s].ready) (void)hipEventDestroy(m->sh[s].ready);
  for (hg_ctx *c : m->ctx) hg_ctx_destroy(c);
  delete m;
}

extern "C" int hg_multi_size(const hg_multi *m) { return m ? (int)m->ctx.size() : 0; }
extern "C" hg_ctx *hg_multi_ctx(hg_multi *m, int shard) {
  return (m && shard >= 0 && shard < (int)m->ctx.size()) ? m->ctx[shard] : nullptr;
}
extern "C" const char *hg_multi_last_error(const hg_multi *m) { return m ? m->err.c_str() : hg_last_error(nullptr); }

// ---- sketch: independent units, no exchange ---------------------------------------------------------------
extern "C" hg_status hg_sketch_batch_multi(hg_multi *m, const uint8_t *const *seqs, const size_t *lens, size_t n,
                                           const hg_sketch_params *p, int16_t *hv_out, int32_t *norm2_out,
                                           uint32_t *nhash_out) {
  if (!m) return HG_ERR_INVALID;
  if (!p) return mfail(m, HG_ERR_INVALID, "params == NULL");
  if (n == 0) return HG_OK;
  if (!seqs || !lens || !hv_out || !norm2_out || !nhash_out) return mfail(m, HG_ERR_INVALID, "NULL argument");
  const int ns = (int)m->ctx.size();
  return for_each_shard(m, [&](int s) -> hg_status {
    size_t lo, hi;
    hg_shard_range(n, s, ns, &lo, &hi);
    if (hi == lo) return HG_OK;
    return hg_sketch_batch(m->ctx[s], seqs + lo, lens + lo, hi - lo, p, hv_out + lo * (size_t)p->hv_d, norm2_out + lo,
                           nhash_out + lo);
  });
}

// ---- dist ------------------------------------------------------------------------------------------------------
namespace {

struct DistPlan {
  std::vector<size_t> rlo, rhi;  // reference rows owned (published) by each shard
  std::vector<size_t> clo, chi;  // query columns computed by each shard (global indices)
  size_t R = 0, Q = 0;
};

// d_ref / d_rn: shard s's reference rows on its device.  d_qry == nullptr: all-vs-all on the gathered matrix.
hg_status dist_core(hg_multi *m, const int16_t *const *d_ref, const int32_t *const *d_rn, const int16_t *const *d_qry,
                    const int32_t *const *d_qn, const DistPlan &pl, uint32_t hv_d, uint32_t ksize, int symmetric,
                    float ani_th, hg_ani_hit *out, size_t cap, size_t *n_out) {
  const int ns = (int)m->ctx.size();
  const size_t row_bytes = (size_t)hv_d * sizeof(int16_t);
  // publish: "my reference rows are complete" on every owner's stream
  for (int s = 0; s < ns; ++s) {
    if (hipSetDevice(m->dev[s]) != hipSuccess || hipEventRecord(m->sh[s].ready, m->ctx[s]->stream) != hipSuccess)
      return mfail(m, HG_ERR_HIP, "hg_dist_multi: event record failed");
  }
  std::vector<size_t> found(ns, 0), caps(ns, 0);
  hg_status st = for_each_shard(m, [&](int s) -> hg_status {
    hg_ctx *c = m->ctx[s];
    hg_multi::Shard &x = m->sh[s];
    const size_t qn_rows = pl.chi[s] - pl.clo[s];
    if (qn_rows == 0) return HG_OK;
    HG_HIP(c, hipSetDevice(m->dev[s]));
    hg_status e;
    if ((e = hg_ensure(c, x.ref_all, pl.R * row_bytes + 64)) != HG_OK) return e;
    if ((e = hg_ensure(c, x.n2_all, pl.R * sizeof(int32_t) + 64)) != HG_OK) return e;
    auto *g_hv = static_cast<int16_t *>(x.ref_all.p);
    auto *g_n2 = static_cast<int32_t *>(x.n2_all.p);
    // all-gather by direct pulls: one copy per peer block, all queued at once on this shard's stream
    for (int t = 0; t < ns; ++t) {
      const size_t rows = pl.rhi[t] - pl.rlo[t];
      if (!rows) continue;
      if (t != s) HG_HIP(c, hipStreamWaitEvent(c->stream, m->sh[t].ready, 0));
      HG_HIP(c, peer_copy(m, s, g_hv + pl.rlo[t] * (size_t)hv_d, t, d_ref[t], rows * row_bytes));
      HG_HIP(c, peer_copy(m, s, g_n2 + pl.rlo[t], t, d_rn[t], rows * sizeof(int32_t)));
    }
    const int16_t *q_hv = d_qry ? d_qry[s] : g_hv + pl.clo[s] * (size_t)hv_d;
    const int32_t *q_n2 = d_qry ? d_qn[s] : g_n2 + pl.clo[s];
    // capacity of this shard's list: its share of the caller's capacity can be exceeded by a skewed hit
    // distribution, so it gets the whole `cap`, bounded by its pair count
    const unsigned __int128 pairs = (unsigned __int128)pl.R * qn_rows;
    caps[s] = (size_t)std::min<unsigned __int128>(pairs, cap);
    if ((e = hg_ensure(c, x.hits, caps[s] * sizeof(hg_ani_hit) + 64)) != HG_OK) return e;
    return hg_dist_block_dev(c, g_hv, g_n2, pl.R, 0, q_hv, q_n2, qn_rows, pl.clo[s], hv_d, ksize, symmetric, ani_th,
                             static_cast<hg_ani_hit *>(x.hits.p), caps[s], &found[s]);
  });
  size_t total = 0;
  for (int s = 0; s < ns; ++s) total += found[s];
  if (n_out) *n_out = total;
  if (st != HG_OK && st != HG_ERR_CAPACITY) return st;
  if (total > cap) return mfail(m, HG_ERR_CAPACITY, "hit buffer too small");
  // merge: shard lists back to back (hit order is unspecified by contract)
  size_t at = 0;
  for (int s = 0; s < ns; ++s) {
    if (!found[s]) continue;
    hg_ctx *c = m->ctx[s];
    HG_HIP(c, hipSetDevice(m->dev[s]));
    HG_HIP(c, hipMemcpyAsync(out + at, m->sh[s].hits.p, found[s] * sizeof(hg_ani_hit), hipMemcpyDeviceToHost, c->stream));
    at += found[s];
  }
  for (int s = 0; s < ns; ++s) {
    HG_HIP(m->ctx[s], hipSetDevice(m->dev[s]));
    HG_HIP(m->ctx[s], hipStreamSynchronize(m->ctx[s]->stream));
  }
  return HG_OK;
}

// column ranges of the all-vs-all case: by pair count under `symmetric` (column j pairs with j rows)
void all_vs_all_columns(size_t n, int ns, int symmetric, DistPlan &pl) {
  pl.clo.resize(ns), pl.chi.resize(ns);
  for (int s = 0; s < ns; ++s) {
    if (symmetric) {
      pl.clo[s] = (size_t)std::llround((double)n * std::sqrt((double)s / ns));
      pl.chi[s] = s + 1 == ns ? n : (size_t)std::llround((double)n * std::sqrt((double)(s + 1) / ns));
    } else {
      hg_shard_range(n, s, ns, &pl.clo[s], &pl.chi[s]);
    }
  }
}

}  // namespace

extern "C" hg_status hg_dist_multi_dev(hg_multi *m, const int16_t *const *d_ref_hv, const int32_t *const *d_ref_norm2,
                                       const size_t *ref_rows, const int16_t *const *d_qry_hv,
                                       const int32_t *const *d_qry_norm2, const size_t *qry_rows, uint32_t hv_d,
                                       uint32_t ksize, int symmetric, float ani_th, hg_ani_hit *out, size_t cap,
                                       size_t *n_out) {
  if (!m) return HG_ERR_INVALID;
  if (!n_out) return mfail(m, HG_ERR_INVALID, "n_out == NULL");
  *n_out = 0;
  if (!d_ref_hv || !d_ref_norm2 || !ref_rows || (cap && !out)) return mfail(m, HG_ERR_INVALID, "NULL argument");
  if (d_qry_hv && (!d_qry_norm2 || !qry_rows)) return mfail(m, HG_ERR_INVALID, "query shards need norms and row counts");
  const int ns = (int)m->ctx.size();
  DistPlan pl;
  pl.rlo.resize(ns), pl.rhi.resize(ns);
  for (int s = 0; s < ns; ++s) {
    pl.rlo[s] = pl.R, pl.R += ref_rows[s], pl.rhi[s] = pl.R;
    if (ref_rows[s] && (!d_ref_hv[s] || !d_ref_norm2[s])) return mfail(m, HG_ERR_INVALID, "NULL reference shard");
  }
  if (d_qry_hv) {
    pl.clo.resize(ns), pl.chi.resize(ns);
    for (int s = 0; s < ns; ++s) {
      pl.clo[s] = pl.Q, pl.Q += qry_rows[s], pl.chi[s] = pl.Q;
      if (qry_rows[s] && (!d_qry_hv[s] || !d_qry_norm2[s])) return mfail(m, HG_ERR_INVALID, "NULL query shard");
    }
  } else {
    pl.Q = pl.R;
    all_vs_all_columns(pl.R, ns, symmetric, pl);
  }
  if (pl.R == 0 || pl.Q == 0) return HG_OK;
  return dist_core(m, d_ref_hv, d_ref_norm2, d_qry_hv, d_qry_norm2, pl, hv_d, ksize, symmetric, ani_th, out, cap, n_out);
}

extern "C" hg_status hg_dist_multi(hg_multi *m, const int16_t *ref_hv, const int32_t *ref_norm2, size_t R,
                                   const int16_t *qry_hv, const int32_t *qry_norm2, size_t Q, uint32_t hv_d,
                                   uint32_t ksize, int symmetric, float ani_th, hg_ani_hit *out, size_t cap,
                                   size_t *n_out) {
  if (!m) return HG_ERR_INVALID;
  if (!n_out) return mfail(m, HG_ERR_INVALID, "n_out == NULL");
  *n_out = 0;
  if (R == 0 || Q == 0) return HG_OK;
  if (!ref_hv || !ref_norm2 || !qry_hv || !qry_norm2 || (cap && !out)) return mfail(m, HG_ERR_INVALID, "NULL argument");
  if (hv_d == 0) return mfail(m, HG_ERR_INVALID, "hv_d == 0");
  const int ns = (int)m->ctx.size();
  const bool same = ref_hv == qry_hv && ref_norm2 == qry_norm2 && R == Q;
  const size_t row_bytes = (size_t)hv_d * sizeof(int16_t);
  DistPlan pl;
  pl.R = R, pl.Q = Q;
  pl.rlo.resize(ns), pl.rhi.resize(ns);
  for (int s = 0; s < ns; ++s) hg_shard_range(R, s, ns, &pl.rlo[s], &pl.rhi[s]);
  if (same) {
    all_vs_all_columns(R, ns, symmetric, pl);
  } else {
    pl.clo.resize(ns), pl.chi.resize(ns);
    for (int s = 0; s < ns; ++s) hg_shard_range(Q, s, ns, &pl.clo[s], &pl.chi[s]);
  }
  // upload: every reference row crosses PCIe once, to the GPU that publishes it; query rows go to their shard
  std::vector<const int16_t *> d_ref(ns), d_qry(ns);
  std::vector<const int32_t *> d_rn(ns), d_qn(ns);
  hg_status st = for_each_shard(m, [&](int s) -> hg_status {
    hg_ctx *c = m->ctx[s];
    hg_multi::Shard &x = m->sh[s];
    HG_HIP(c, hipSetDevice(m->dev[s]));
    hg_status e;
    const size_t rr = pl.rhi[s] - pl.rlo[s];
    if ((e = hg_ensure(c, x.mine, rr * row_bytes + 64)) != HG_OK) return e;
    if ((e = hg_ensure(c, x.mine_n2, rr * sizeof(int32_t) + 64)) != HG_OK) return e;
    if (rr) {
      HG_HIP(c, hipMemcpyAsync(x.mine.p, ref_hv + pl.rlo[s] * (size_t)hv_d, rr * row_bytes, hipMemcpyHostToDevice, c->stream));
      HG_HIP(c, hipMemcpyAsync(x.mine_n2.p, ref_norm2 + pl.rlo[s], rr * sizeof(int32_t), hipMemcpyHostToDevice, c->stream));
    }
    d_ref[s] = static_cast<const int16_t *>(x.mine.p), d_rn[s] = static_cast<const int32_t *>(x.mine_n2.p);
    if (!same) {
      const size_t qq = pl.chi[s] - pl.clo[s];
      if ((e = hg_ensure(c, x.qry, qq * row_bytes + 64)) != HG_OK) return e;
      if ((e = hg_ensure(c, x.qry_n2, qq * sizeof(int32_t) + 64)) != HG_OK) return e;
      if (qq) {
        HG_HIP(c, hipMemcpyAsync(x.qry.p, qry_hv + pl.clo[s] * (size_t)hv_d, qq * row_bytes, hipMemcpyHostToDevice, c->stream));
        HG_HIP(c, hipMemcpyAsync(x.qry_n2.p, qry_norm2 + pl.clo[s], qq * sizeof(int32_t), hipMemcpyHostToDevice, c->stream));
      }
      d_qry[s] = static_cast<const int16_t *>(x.qry.p), d_qn[s] = static_cast<const int32_t *>(x.qry_n2.p);
    }
    return HG_OK;
  });
  if (st != HG_OK) return st;
  return dist_core(m, d_ref.data(), d_rn.data(), same ? nullptr : d_qry.data(), same ? nullptr : d_qn.data(), pl, hv_d,
                   ksize, symmetric, ani_th, out, cap, n_out);
}

// ---- bit-packed database search: references sharded, queries broadcast, hits merged ----------------------------
extern "C" hg_status hg_hamming_search_multi(hg_multi *m, const uint32_t *ref_bits, size_t R, const uint32_t *qry_bits,
                                             size_t Q, uint32_t hv_d, uint32_t max_dist, hg_ham_hit *out, size_t cap,
                                             size_t *n_out) {
  if (!m) return HG_ERR_INVALID;
  if (!n_out) return mfail(m, HG_ERR_INVALID, "n_out == NULL");
  *n_out = 0;
  if (R == 0 || Q == 0) return HG_OK;
  if (!ref_bits || !qry_bits || (cap && !out)) return mfail(m, HG_ERR_INVALID, "NULL argument");
  const int ns = (int)m->ctx.size();
  const size_t words = (hv_d + 31) / 32, row_bytes = words * sizeof(uint32_t);
  std::vector<size_t> found(ns, 0), caps(ns, 0);
  hg_status st = for_each_shard(m, [&](int s) -> hg_status {
    size_t lo, hi;
    hg_shard_range(R, s, ns, &lo, &hi);
    if (hi == lo) return HG_OK;
    hg_ctx *c = m->ctx[s];
    hg_multi::Shard &x = m->sh[s];
    HG_HIP(c, hipSetDevice(m->dev[s]));
    hg_status e;
    if ((e = hg_ensure(c, x.mine, (hi - lo) * row_bytes + 64)) != HG_OK) return e;
    if ((e = hg_ensure(c, x.qry, Q * row_bytes + 64)) != HG_OK) return e;
    const unsigned __int128 pairs = (unsigned __int128)(hi - lo) * Q;
    caps[s] = (size_t)std::min<unsigned __int128>(pairs, cap);
    if ((e = hg_ensure(c, x.hits, caps[s] * sizeof(hg_ham_hit) + 64)) != HG_OK) return e;
    HG_HIP(c, hipMemcpyAsync(x.mine.p, ref_bits + lo * words, (hi - lo) * row_bytes, hipMemcpyHostToDevice, c->stream));
    HG_HIP(c, hipMemcpyAsync(x.qry.p, qry_bits, Q * row_bytes, hipMemcpyHostToDevice, c->stream));  // broadcast: one PCIe link per GPU
    return hg_hamming_search_block_dev(c, static_cast<const uint32_t *>(x.mine.p), hi - lo, lo,
                                       static_cast<const uint32_t *>(x.qry.p), Q, 0, hv_d, max_dist,
                                       static_cast<hg_ham_hit *>(x.hits.p), caps[s], &found[s]);
  });
  size_t total = 0;
  for (int s = 0; s < ns; ++s) total += found[s];
  *n_out = total;
  if (st != HG_OK && st != HG_ERR_CAPACITY) return st;
  if (total > cap) return mfail(m, HG_ERR_CAPACITY, "hit buffer too small");
  size_t at = 0;
  for (int s = 0; s < ns; ++s) {
    if (!found[s]) continue;
    hg_ctx *c = m->ctx[s];
    HG_HIP(c, hipSetDevice(m->dev[s]));
    HG_HIP(c, hipMemcpyAsync(out + at, m->sh[s].hits.p, found[s] * sizeof(hg_ham_hit), hipMemcpyDeviceToHost, c->stream));
    at += found[s];
  }
  for (int s = 0; s < ns; ++s) {
    HG_HIP(m->ctx[s], hipSetDevice(m->dev[s]));
    HG_HIP(m->ctx[s], hipStreamSynchronize(m->ctx[s]->stream));
  }
  return HG_OK;
}
