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
//
// The same exchange through RCCL (hg_multi_set_gather(m, HG_GATHER_RCCL); SURVEY.md 8(e): "one ncclAllGather of
// the R x D i16 ref matrix + R i32 norms"): one communicator per shard from ncclCommInitAll, the collective queued
// on every shard's stream inside one ncclGroupStart/End -- ncclAllGather when the row blocks have equal sizes,
// one ncclBroadcast per owner otherwise (the all-gather-v idiom).  librccl is opened with dlopen the first time
// the mode is selected, so the library carries no link-time dependency on it (the Makefile links -ldl for dlopen itself:
// glibc < 2.34 keeps it in libdl), and no build-time one either: without the rccl-dev header the handful of
// declarations used here are restated from the public NCCL API.
#include <dlfcn.h>
#if __has_include(<rccl/rccl.h>)
#include <rccl/rccl.h>
#else
extern "C" {
typedef struct ncclComm *ncclComm_t;
typedef enum { ncclSuccess = 0, ncclUnhandledCudaError = 1, ncclSystemError = 2, ncclInternalError = 3 } ncclResult_t;
typedef enum { ncclInt8 = 0, ncclUint8 = 1 } ncclDataType_t;
ncclResult_t ncclCommInitAll(ncclComm_t *comm, int ndev, const int *devlist);
ncclResult_t ncclCommDestroy(ncclComm_t comm);
ncclResult_t ncclAllGather(const void *sendbuff, void *recvbuff, size_t sendcount, ncclDataType_t datatype, ncclComm_t comm,
                           hipStream_t stream);
ncclResult_t ncclBroadcast(const void *sendbuff, void *recvbuff, size_t count, ncclDataType_t datatype, int root,
                           ncclComm_t comm, hipStream_t stream);
ncclResult_t ncclGroupStart(void);
ncclResult_t ncclGroupEnd(void);
const char *ncclGetErrorString(ncclResult_t result);
ncclResult_t ncclGetVersion(int *version);
}
#endif

#include <algorithm>
#include <cmath>
#include <cstring>
#include <mutex>
#include <thread>
#include <vector>

#include "hg_internal.h"

namespace {
struct RcclApi {
  decltype(&ncclCommInitAll) CommInitAll = nullptr;
  decltype(&ncclCommDestroy) CommDestroy = nullptr;
  decltype(&ncclAllGather) AllGather = nullptr;
  decltype(&ncclBroadcast) Broadcast = nullptr;
  decltype(&ncclGroupStart) GroupStart = nullptr;
  decltype(&ncclGroupEnd) GroupEnd = nullptr;
  decltype(&ncclGetErrorString) GetErrorString = nullptr;
  decltype(&ncclGetVersion) GetVersion = nullptr;
  std::string why;  // why it is unusable, if it is
  bool ok = false;
};
const RcclApi &rccl() {
  static RcclApi api;
  static std::once_flag once;
  std::call_once(once, [] {
    void *h = nullptr;
    // a copy that is already in the process (PyTorch ships its own) is reused through the soname lookup
    for (const char *name : {"librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1"})
      if ((h = dlopen(name, RTLD_NOW | RTLD_GLOBAL))) break;
    if (!h) {
      const char *e = dlerror();
      api.why = std::string("librccl not loadable: ") + (e ? e : "?");
      return;
    }
    bool all = true;
    auto sym = [&](const char *n) {
      void *p = dlsym(h, n);
      if (!p) all = false, api.why = std::string("librccl lacks ") + n;
      return p;
    };
    api.CommInitAll = reinterpret_cast<decltype(api.CommInitAll)>(sym("ncclCommInitAll"));
    api.CommDestroy = reinterpret_cast<decltype(api.CommDestroy)>(sym("ncclCommDestroy"));
    api.AllGather = reinterpret_cast<decltype(api.AllGather)>(sym("ncclAllGather"));
    api.Broadcast = reinterpret_cast<decltype(api.Broadcast)>(sym("ncclBroadcast"));
    api.GroupStart = reinterpret_cast<decltype(api.GroupStart)>(sym("ncclGroupStart"));
    api.GroupEnd = reinterpret_cast<decltype(api.GroupEnd)>(sym("ncclGroupEnd"));
    api.GetErrorString = reinterpret_cast<decltype(api.GetErrorString)>(sym("ncclGetErrorString"));
    api.GetVersion = reinterpret_cast<decltype(api.GetVersion)>(sym("ncclGetVersion"));
    api.ok = all;
  });
  return api;
}
}  // namespace

struct hg_multi {
  std::vector<hg_ctx *> ctx;
  std::vector<int> dev;
  std::string err;
  struct Shard {
    hg_ctx::Buf ref_all, n2_all;  // gathered reference matrix + norms (every shard holds all rows)
    hg_ctx::Buf mine, mine_n2;    // this shard's uploaded reference rows (host entry points)
    hg_ctx::Buf qry, qry_n2;      // this shard's query rows (host entry points)
    hg_ctx::Buf hits;             // per-shard hit list
    // the exchange of PREPARED operands (dist_core_ops): this shard's own rows as byte operands + control records + flag
    // word, and what it gathers from all owners
    hg_ctx::Buf ops_mine, meta_mine, flag_mine, ops_all, meta_all, flags_all;
    hipEvent_t ready = nullptr;   // "this shard's published rows are complete"
  };
  std::vector<Shard> sh;
  int gather = HG_GATHER_PEER;
  std::vector<ncclComm_t> comm;  // one per shard once HG_GATHER_RCCL was selected
  std::string peer_report;       // what hipDeviceEnablePeerAccess said at creation
  std::string gather_report;     // how the last exchange step ran
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
  // direct xGMI access between distinct devices (hipMemcpyPeerAsync stages through the host without it).  The
  // outcome is kept for the caller (hg_multi_peer_report): a pair without peer access still works, at PCIe speed.
  int pairs = 0, enabled = 0;
  std::string failed;
  for (int a = 0; a < n; ++a)
    for (int b = 0; b < n; ++b) {
      if (m->dev[a] == m->dev[b]) continue;
      bool seen = false;  // repeated ids: report every ordered device pair once
      for (int a2 = 0; a2 <= a && !seen; ++a2)
        for (int b2 = 0; b2 < (a2 == a ? b : n) && !seen; ++b2)
          seen = m->dev[a2] == m->dev[a] && m->dev[b2] == m->dev[b];
      if (seen) continue;
      ++pairs;
      int can = 0;
      hipError_t e = hipDeviceCanAccessPeer(&can, m->dev[a], m->dev[b]);
      if (e == hipSuccess && can) {
        (void)hipSetDevice(m->dev[a]);
        e = hipDeviceEnablePeerAccess(m->dev[b], 0);
        if (e == hipErrorPeerAccessAlreadyEnabled) e = hipSuccess, (void)hipGetLastError();
      } else if (e == hipSuccess) {
        e = hipErrorPeerAccessUnsupported;
      }
      if (e == hipSuccess) {
        ++enabled;
      } else {
        (void)hipGetLastError();
        failed += " " + std::to_string(m->dev[a]) + "->" + std::to_string(m->dev[b]) + " (" + hipGetErrorName(e) + ")";
      }
    }
  m->peer_report = "peer access enabled for " + std::to_string(enabled) + " of " + std::to_string(pairs) +
                   " ordered device pairs" + (failed.empty() ? std::string() : "; host-staged copies for:" + failed);
  *out = m;
  return HG_OK;
}

extern "C" const char *hg_multi_peer_report(const hg_multi *m) { return m ? m->peer_report.c_str() : ""; }
extern "C" const char *hg_multi_gather_report(const hg_multi *m) { return m ? m->gather_report.c_str() : ""; }
extern "C" int hg_multi_gather_mode(const hg_multi *m) { return m ? m->gather : -1; }

extern "C" hg_status hg_multi_set_gather(hg_multi *m, int mode) {
  if (!m) return HG_ERR_INVALID;
  if (mode == HG_GATHER_PEER) {
    m->gather = mode;
    return HG_OK;
  }
  if (mode != HG_GATHER_RCCL) return mfail(m, HG_ERR_INVALID, "gather mode: HG_GATHER_PEER or HG_GATHER_RCCL");
  const int n = (int)m->ctx.size();
  for (int a = 0; a < n; ++a)
    for (int b = a + 1; b < n; ++b)
      if (m->dev[a] == m->dev[b])
        return mfail(m, HG_ERR_UNSUPPORTED, "HG_GATHER_RCCL needs distinct devices (one communicator rank per GPU)");
  const RcclApi &r = rccl();
  if (!r.ok) return mfail(m, HG_ERR_UNSUPPORTED, r.why);
  if (m->comm.empty()) {
    for (int s = 0; s < n; ++s) (void)hipSetDevice(m->dev[s]), (void)hipStreamSynchronize(m->ctx[s]->stream);
    std::vector<ncclComm_t> comm(n, nullptr);
    const ncclResult_t e = r.CommInitAll(comm.data(), n, m->dev.data());
    if (e != ncclSuccess) return mfail(m, HG_ERR_HIP, std::string("ncclCommInitAll: ") + r.GetErrorString(e));
    m->comm = comm;
  }
  m->gather = mode;
  return HG_OK;
}

extern "C" void hg_multi_destroy(hg_multi *m) {
  if (!m) return;
  for (size_t s = 0; s < m->ctx.size(); ++s) {
    (void)hipSetDevice(m->dev[s]);
    (void)hipStreamSynchronize(m->ctx[s]->stream);
  }
  for (ncclComm_t c : m->comm)
    if (c) (void)rccl().CommDestroy(c);
  for (size_t s = 0; s < m->ctx.size(); ++s) {
    (void)hipSetDevice(m->dev[s]);
    hg_multi::Shard &x = m->sh[s];
    for (hg_ctx::Buf *b : {&x.ref_all, &x.n2_all, &x.mine, &x.mine_n2, &x.qry, &x.qry_n2, &x.hits, &x.ops_mine, &x.meta_mine,
                           &x.flag_mine, &x.ops_all, &x.meta_all, &x.flags_all})
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

// every shard's stream idle: no copy queued by a call may outlive it (a peer pull still reading another shard's
// `mine` buffer while the caller's retry re-uploads into it)
void drain(hg_multi *m) {
  for (size_t s = 0; s < m->ctx.size(); ++s)
    if (hipSetDevice(m->dev[s]) == hipSuccess) (void)hipStreamSynchronize(m->ctx[s]->stream);
}
struct DrainOnExit {
  hg_multi *m;
  ~DrainOnExit() { drain(m); }
};

// the exchange step: every shard's gathered matrix g_hv[s] / g_n2[s] receives all owners' row blocks
hg_status gather_refs(hg_multi *m, const int16_t *const *d_ref, const int32_t *const *d_rn, const DistPlan &pl,
                      uint32_t hv_d, const std::vector<int16_t *> &g_hv, const std::vector<int32_t *> &g_n2) {
  const int ns = (int)m->ctx.size();
  const size_t row_bytes = (size_t)hv_d * sizeof(int16_t);
  if (m->gather == HG_GATHER_RCCL && !m->comm.empty()) {
    const RcclApi &r = rccl();
    bool equal = true;
    for (int t = 1; t < ns; ++t) equal = equal && (pl.rhi[t] - pl.rlo[t]) == (pl.rhi[0] - pl.rlo[0]);
    ncclResult_t e = r.GroupStart();
    for (int s = 0; s < ns && e == ncclSuccess; ++s) {
      if (!g_hv[s]) continue;  // (a shard without query rows still has to take part: see the caller)
      if (hipSetDevice(m->dev[s]) != hipSuccess) {  // (no return inside the group: GroupEnd below closes it on every path)
        e = ncclSystemError;
        break;
      }
      hipStream_t st = m->ctx[s]->stream;
      if (equal) {
        const size_t rows = pl.rhi[s] - pl.rlo[s];
        e = r.AllGather(d_ref[s], g_hv[s], rows * row_bytes, ncclUint8, m->comm[s], st);
        if (e == ncclSuccess) e = r.AllGather(d_rn[s], g_n2[s], rows * sizeof(int32_t), ncclUint8, m->comm[s], st);
      } else {
        for (int t = 0; t < ns && e == ncclSuccess; ++t) {
          const size_t rows = pl.rhi[t] - pl.rlo[t];
          if (!rows) continue;
          e = r.Broadcast(d_ref[s], g_hv[s] + pl.rlo[t] * (size_t)hv_d, rows * row_bytes, ncclUint8, t, m->comm[s], st);
          if (e == ncclSuccess)
            e = r.Broadcast(d_rn[s], g_n2[s] + pl.rlo[t], rows * sizeof(int32_t), ncclUint8, t, m->comm[s], st);
        }
      }
    }
    const ncclResult_t e2 = r.GroupEnd();
    if (e == ncclSuccess) e = e2;
    if (e != ncclSuccess) return mfail(m, HG_ERR_HIP, std::string("RCCL all-gather: ") + r.GetErrorString(e));
    int ver = 0;
    (void)r.GetVersion(&ver);
    m->gather_report = std::string("rccl ") + (equal ? "ncclAllGather" : "grouped ncclBroadcast") + " over " +
                       std::to_string(ns) + " ranks, " + std::to_string(pl.R * row_bytes) + " B, version " + std::to_string(ver);
    return HG_OK;
  }
  // direct pulls: one copy per peer block, all queued at once on the puller's stream
  for (int s = 0; s < ns; ++s) {
    if (hipSetDevice(m->dev[s]) != hipSuccess || hipEventRecord(m->sh[s].ready, m->ctx[s]->stream) != hipSuccess)
      return mfail(m, HG_ERR_HIP, "hg_dist_multi: event record failed");
  }
  size_t peer_bytes = 0;
  for (int s = 0; s < ns; ++s) {
    if (!g_hv[s]) continue;
    hg_ctx *c = m->ctx[s];
    HG_HIP(c, hipSetDevice(m->dev[s]));
    for (int t = 0; t < ns; ++t) {
      const size_t rows = pl.rhi[t] - pl.rlo[t];
      if (!rows) continue;
      if (t != s) HG_HIP(c, hipStreamWaitEvent(c->stream, m->sh[t].ready, 0));
      HG_HIP(c, peer_copy(m, s, g_hv[s] + pl.rlo[t] * (size_t)hv_d, t, d_ref[t], rows * row_bytes));
      HG_HIP(c, peer_copy(m, s, g_n2[s] + pl.rlo[t], t, d_rn[t], rows * sizeof(int32_t)));
      if (m->dev[s] != m->dev[t]) peer_bytes += rows * row_bytes;
    }
  }
  m->gather_report = "peer pulls (hipMemcpyPeerAsync), " + std::to_string(peer_bytes) + " B between distinct devices; " +
                     m->peer_report;
  return HG_OK;
}

// shard lists back to back into the caller's buffer (hit order is unspecified by contract)
hg_status merge_hits(hg_multi *m, const std::vector<size_t> &found, hg_status st, hg_ani_hit *out, size_t cap, size_t *n_out) {
  const int ns = (int)m->ctx.size();
  size_t total = 0;
  for (int s = 0; s < ns; ++s) total += found[s];
  if (n_out) *n_out = total;
  if (st != HG_OK && st != HG_ERR_CAPACITY) return st;
  if (total > cap) return mfail(m, HG_ERR_CAPACITY, "hit buffer too small");
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

// The exchange of PREPARED operands: every shard converts its own reference rows to centred byte operands + 72-byte
// control records once (hg_dist_prep_ops_dev), and the peers pull those -- 4.3 KB per row at D = 4096 instead of 8 KB of
// i16, and no shard repeats another shard's prepass.  In the all-vs-all case a shard additionally pulls the i16 rows of
// its QUERY range that it does not own itself (under `symmetric` the column ranges are balanced by pair count and do not
// coincide with the row blocks).  Returns HG_ERR_INEXACT, with nothing reported, when an owner's or a query side's rows
// do not fit the byte scheme: dist_core then runs the i16 exchange.
hg_status dist_core_ops(hg_multi *m, const int16_t *const *d_ref, const int32_t *const *d_rn, const int16_t *const *d_qry,
                        const int32_t *const *d_qn, const DistPlan &pl, uint32_t hv_d, uint32_t ksize, int symmetric,
                        float ani_th, hg_ani_hit *out, size_t cap, size_t *n_out) {
  const int ns = (int)m->ctx.size();
  const size_t rb = hg_dist_ops_row_bytes(hv_d), mb = hg_dist_ops_meta_bytes(), row16 = (size_t)hv_d * sizeof(int16_t);
  std::vector<size_t> found(ns, 0), caps(ns, 0);
  const bool rccl_mode = m->gather == HG_GATHER_RCCL && !m->comm.empty();
  // phase 1 (one thread per shard): workspaces, and the shard's own rows prepared on its stream
  hg_status st = for_each_shard(m, [&](int s) -> hg_status {
    hg_ctx *c = m->ctx[s];
    hg_multi::Shard &x = m->sh[s];
    const size_t rows = pl.rhi[s] - pl.rlo[s], qn_rows = pl.chi[s] - pl.clo[s];
    HG_HIP(c, hipSetDevice(m->dev[s]));
    hg_status e;
    if ((e = hg_ensure(c, x.ops_mine, rows * rb + 64)) != HG_OK) return e;
    if ((e = hg_ensure(c, x.meta_mine, rows * mb + 64)) != HG_OK) return e;
    if ((e = hg_ensure(c, x.flag_mine, 64)) != HG_OK) return e;
    if (qn_rows || rccl_mode) {
      if ((e = hg_ensure(c, x.ops_all, hg_dist_ops_padded_rows(pl.R) * rb + 64)) != HG_OK) return e;
      if ((e = hg_ensure(c, x.meta_all, pl.R * mb + 64)) != HG_OK) return e;
      if ((e = hg_ensure(c, x.n2_all, pl.R * sizeof(int32_t) + 64)) != HG_OK) return e;
      if ((e = hg_ensure(c, x.flags_all, (size_t)ns * sizeof(uint32_t) + 64)) != HG_OK) return e;
    }
    if (!d_qry && qn_rows) {  // all-vs-all: this shard's query rows, assembled below
      if ((e = hg_ensure(c, x.qry, qn_rows * row16 + 64)) != HG_OK) return e;
      if ((e = hg_ensure(c, x.qry_n2, qn_rows * sizeof(int32_t) + 64)) != HG_OK) return e;
    }
    const unsigned __int128 pairs = (unsigned __int128)pl.R * qn_rows;
    caps[s] = (size_t)std::min<unsigned __int128>(pairs, cap);
    if ((e = hg_ensure(c, x.hits, caps[s] * sizeof(hg_ani_hit) + 64)) != HG_OK) return e;
    HG_HIP(c, hipMemsetAsync(x.flag_mine.p, 0, 16, c->stream));
    if (rows)
      return hg_dist_prep_ops_dev(c, d_ref[s], rows, hv_d, static_cast<uint8_t *>(x.ops_mine.p), static_cast<uint8_t *>(x.meta_mine.p),
                                  static_cast<uint32_t *>(x.flag_mine.p));
    return HG_OK;
  });
  if (st != HG_OK) return st;
  // phase 2 (this thread): the exchange, queued on the pullers' streams behind the owners' "ready" events
  size_t peer_bytes = 0;
  bool equal = true;
  for (int t = 1; t < ns; ++t) equal = equal && (pl.rhi[t] - pl.rlo[t]) == (pl.rhi[0] - pl.rlo[0]);
  if (rccl_mode) {
    const RcclApi &r = rccl();
    ncclResult_t e = r.GroupStart();
    for (int s = 0; s < ns && e == ncclSuccess; ++s) {
      if (hipSetDevice(m->dev[s]) != hipSuccess) {
        e = ncclSystemError;
        break;
      }
      hg_multi::Shard &x = m->sh[s];
      hipStream_t stq = m->ctx[s]->stream;
      auto *oa = static_cast<uint8_t *>(x.ops_all.p), *ma = static_cast<uint8_t *>(x.meta_all.p);
      auto *na = static_cast<int32_t *>(x.n2_all.p);
      auto *fa = static_cast<uint32_t *>(x.flags_all.p);
      if (equal) {
        const size_t rows = pl.rhi[s] - pl.rlo[s];
        e = r.AllGather(x.ops_mine.p, oa, rows * rb, ncclUint8, m->comm[s], stq);
        if (e == ncclSuccess) e = r.AllGather(x.meta_mine.p, ma, rows * mb, ncclUint8, m->comm[s], stq);
        if (e == ncclSuccess) e = r.AllGather(d_rn[s], na, rows * sizeof(int32_t), ncclUint8, m->comm[s], stq);
      } else {
        for (int t = 0; t < ns && e == ncclSuccess; ++t) {
          const size_t rows = pl.rhi[t] - pl.rlo[t];
          if (!rows) continue;
          e = r.Broadcast(x.ops_mine.p, oa + pl.rlo[t] * rb, rows * rb, ncclUint8, t, m->comm[s], stq);
          if (e == ncclSuccess) e = r.Broadcast(x.meta_mine.p, ma + pl.rlo[t] * mb, rows * mb, ncclUint8, t, m->comm[s], stq);
          if (e == ncclSuccess) e = r.Broadcast(d_rn[s], na + pl.rlo[t], rows * sizeof(int32_t), ncclUint8, t, m->comm[s], stq);
        }
      }
      if (e == ncclSuccess) e = r.AllGather(x.flag_mine.p, fa, sizeof(uint32_t), ncclUint8, m->comm[s], stq);
    }
    const ncclResult_t e2 = r.GroupEnd();
    if (e == ncclSuccess) e = e2;
    if (e != ncclSuccess) return mfail(m, HG_ERR_HIP, std::string("RCCL all-gather: ") + r.GetErrorString(e));
  }
  for (int s = 0; s < ns; ++s)
    if (hipSetDevice(m->dev[s]) != hipSuccess || hipEventRecord(m->sh[s].ready, m->ctx[s]->stream) != hipSuccess)
      return mfail(m, HG_ERR_HIP, "hg_dist_multi: event record failed");
  for (int s = 0; s < ns; ++s) {
    const size_t qn_rows = pl.chi[s] - pl.clo[s];
    if (!qn_rows) continue;
    hg_ctx *c = m->ctx[s];
    hg_multi::Shard &x = m->sh[s];
    HG_HIP(c, hipSetDevice(m->dev[s]));
    for (int t = 0; t < ns; ++t) {
      const size_t rows = pl.rhi[t] - pl.rlo[t];
      if (!rows) continue;
      if (t != s) HG_HIP(c, hipStreamWaitEvent(c->stream, m->sh[t].ready, 0));
      if (!rccl_mode) {
        HG_HIP(c, peer_copy(m, s, static_cast<uint8_t *>(x.ops_all.p) + pl.rlo[t] * rb, t, m->sh[t].ops_mine.p, rows * rb));
        HG_HIP(c, peer_copy(m, s, static_cast<uint8_t *>(x.meta_all.p) + pl.rlo[t] * mb, t, m->sh[t].meta_mine.p, rows * mb));
        HG_HIP(c, peer_copy(m, s, static_cast<int32_t *>(x.n2_all.p) + pl.rlo[t], t, d_rn[t], rows * sizeof(int32_t)));
        HG_HIP(c, peer_copy(m, s, static_cast<uint32_t *>(x.flags_all.p) + t, t, m->sh[t].flag_mine.p, sizeof(uint32_t)));
        if (m->dev[s] != m->dev[t]) peer_bytes += rows * (rb + mb + 4);
      }
      if (!d_qry) {  // all-vs-all: the part of this shard's query range that owner t holds, as i16 rows
        const size_t lo = std::max(pl.clo[s], pl.rlo[t]), hi = std::min(pl.chi[s], pl.rhi[t]);
        if (lo < hi) {
          HG_HIP(c, peer_copy(m, s, static_cast<int16_t *>(x.qry.p) + (lo - pl.clo[s]) * (size_t)hv_d, t,
                              d_ref[t] + (lo - pl.rlo[t]) * (size_t)hv_d, (hi - lo) * row16));
          HG_HIP(c, peer_copy(m, s, static_cast<int32_t *>(x.qry_n2.p) + (lo - pl.clo[s]), t, d_rn[t] + (lo - pl.rlo[t]),
                              (hi - lo) * sizeof(int32_t)));
          if (m->dev[s] != m->dev[t]) peer_bytes += (hi - lo) * (row16 + 4);
        }
      }
    }
  }
  // phase 3 (one thread per shard): (all refs, as gathered operands) x (this shard's query rows)
  std::vector<hg_status> sst(ns, HG_OK);
  st = for_each_shard(m, [&](int s) -> hg_status {
    hg_ctx *c = m->ctx[s];
    hg_multi::Shard &x = m->sh[s];
    const size_t qn_rows = pl.chi[s] - pl.clo[s];
    if (qn_rows == 0) return HG_OK;
    HG_HIP(c, hipSetDevice(m->dev[s]));
    const int16_t *q_hv = d_qry ? d_qry[s] : static_cast<const int16_t *>(x.qry.p);
    const int32_t *q_n2 = d_qry ? d_qn[s] : static_cast<const int32_t *>(x.qry_n2.p);
    sst[s] = hg_dist_block_ops_dev(c, static_cast<const uint8_t *>(x.ops_all.p), static_cast<const uint8_t *>(x.meta_all.p),
                                   static_cast<const int32_t *>(x.n2_all.p), pl.R, 0, nullptr,
                                   static_cast<const uint32_t *>(x.flags_all.p), (size_t)ns, q_hv, q_n2, qn_rows, pl.clo[s], hv_d,
                                   ksize, symmetric, ani_th, static_cast<hg_ani_hit *>(x.hits.p), caps[s], &found[s]);
    return sst[s] == HG_ERR_INEXACT ? HG_OK : sst[s];
  });
  for (int s = 0; s < ns; ++s)
    if (sst[s] == HG_ERR_INEXACT) return HG_ERR_INEXACT;  // (an owner's veto is seen by every shard; a query side's by its own)
  m->gather_report = (rccl_mode ? std::string("rccl ") + (equal ? "ncclAllGather" : "grouped ncclBroadcast") + " over " + std::to_string(ns) + " ranks"
                                : std::string("peer pulls (hipMemcpyPeerAsync)")) + ", prepared byte operands + control records: " +
                     std::to_string(pl.R * (rb + mb + 4)) + " B per shard instead of " + std::to_string(pl.R * (row16 + 4)) +
                     " B of i16 rows, " + std::to_string(peer_bytes) + " B between distinct devices; " + m->peer_report;
  return merge_hits(m, found, st, out, cap, n_out);
}

// d_ref / d_rn: shard s's reference rows on its device.  d_qry == nullptr: all-vs-all on the gathered matrix.
hg_status dist_core(hg_multi *m, const int16_t *const *d_ref, const int32_t *const *d_rn, const int16_t *const *d_qry,
                    const int32_t *const *d_qn, const DistPlan &pl, uint32_t hv_d, uint32_t ksize, int symmetric,
                    float ani_th, hg_ani_hit *out, size_t cap, size_t *n_out) {
  const int ns = (int)m->ctx.size();
  const size_t row_bytes = (size_t)hv_d * sizeof(int16_t);
  DrainOnExit drain_guard{m};  // every return below leaves all shard streams idle
  // the exchange of prepared operands first (half the bytes, no repeated prepass); a veto -- sketches that do not fit the
  // byte scheme -- or the "f16" hook on shard 0 brings the i16 exchange below
  if (hv_d <= 8192 && hv_d % 8 == 0 && m->ctx[0]->dbg_dist_path != "f16") {
    const hg_status os = dist_core_ops(m, d_ref, d_rn, d_qry, d_qn, pl, hv_d, ksize, symmetric, ani_th, out, cap, n_out);
    if (os != HG_ERR_INEXACT) return os;
    drain(m);
    if (n_out) *n_out = 0;
  }
  std::vector<size_t> found(ns, 0), caps(ns, 0);
  std::vector<int16_t *> g_hv(ns, nullptr);
  std::vector<int32_t *> g_n2(ns, nullptr);
  const bool rccl_mode = m->gather == HG_GATHER_RCCL && !m->comm.empty();
  // phase 1 (one thread per shard): workspaces.  Under RCCL every rank takes part in the collective, so every
  // shard gets a gathered matrix; the peer pulls skip shards that have no query rows.
  hg_status st = for_each_shard(m, [&](int s) -> hg_status {
    hg_ctx *c = m->ctx[s];
    hg_multi::Shard &x = m->sh[s];
    const size_t qn_rows = pl.chi[s] - pl.clo[s];
    if (qn_rows == 0 && !rccl_mode) return HG_OK;
    HG_HIP(c, hipSetDevice(m->dev[s]));
    hg_status e;
    if ((e = hg_ensure(c, x.ref_all, pl.R * row_bytes + 64)) != HG_OK) return e;
    if ((e = hg_ensure(c, x.n2_all, pl.R * sizeof(int32_t) + 64)) != HG_OK) return e;
    g_hv[s] = static_cast<int16_t *>(x.ref_all.p), g_n2[s] = static_cast<int32_t *>(x.n2_all.p);
    // capacity of this shard's list: its share of the caller's capacity can be exceeded by a skewed hit
    // distribution, so it gets the whole `cap`, bounded by its pair count
    const unsigned __int128 pairs = (unsigned __int128)pl.R * qn_rows;
    caps[s] = (size_t)std::min<unsigned __int128>(pairs, cap);
    return hg_ensure(c, x.hits, caps[s] * sizeof(hg_ani_hit) + 64);
  });
  if (st != HG_OK) return st;
  // phase 2 (this thread): the exchange step, queued on the shards' streams
  if ((st = gather_refs(m, d_ref, d_rn, pl, hv_d, g_hv, g_n2)) != HG_OK) return st;
  // phase 3 (one thread per shard): (all refs) x (this shard's query rows)
  st = for_each_shard(m, [&](int s) -> hg_status {
    hg_ctx *c = m->ctx[s];
    const size_t qn_rows = pl.chi[s] - pl.clo[s];
    if (qn_rows == 0) return HG_OK;
    HG_HIP(c, hipSetDevice(m->dev[s]));
    const int16_t *q_hv = d_qry ? d_qry[s] : g_hv[s] + pl.clo[s] * (size_t)hv_d;
    const int32_t *q_n2 = d_qry ? d_qn[s] : g_n2[s] + pl.clo[s];
    return hg_dist_block_dev(c, g_hv[s], g_n2[s], pl.R, 0, q_hv, q_n2, qn_rows, pl.clo[s], hv_d, ksize, symmetric, ani_th,
                             static_cast<hg_ani_hit *>(m->sh[s].hits.p), caps[s], &found[s]);
  });
  return merge_hits(m, found, st, out, cap, n_out);
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
  for (int s = 0; s < ns; ++s) {  // the shards' operands may be the outputs of a sketch step still queued on their ctx
    if (hipSetDevice(m->dev[s]) != hipSuccess) return mfail(m, HG_ERR_HIP, "hipSetDevice");
    const hg_status st = hg_sketch_resolve(m->ctx[s]);
    if (st != HG_OK) return mfail(m, st, hg_last_error(m->ctx[s]));
  }
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
  DrainOnExit drain_guard{m};  // every return below leaves all shard streams idle
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
