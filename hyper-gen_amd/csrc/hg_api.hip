// hg_api.hip -- the C ABI of include/hypergen.h, part 1: context, streams, workspaces, timing brackets, memory helpers.
// (hg_api_sketch.hip: the sketch entry points; hg_api_dist.hip: the dist entry points.)  No CPU fallback lives anywhere:
// every compute entry point runs HIP kernels or fails.
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

static thread_local std::string g_create_err;

hg_status hg_fail(hg_ctx *ctx, hg_status s, const std::string &msg) {
  if (ctx) ctx->err = msg;
  else g_create_err = msg;
  return s;
}

extern "C" const char *hg_status_str(hg_status s) {
  switch (s) {
    case HG_OK: return "ok";
    case HG_ERR_INVALID: return "invalid argument";
    case HG_ERR_NO_DEVICE: return "no usable HIP device";
    case HG_ERR_HIP: return "HIP runtime error";
    case HG_ERR_OOM: return "out of memory";
    case HG_ERR_CAPACITY: return "output capacity too small";
    case HG_ERR_UNSUPPORTED: return "unsupported parameters";
    case HG_ERR_IO: return "I/O error";
    case HG_ERR_INEXACT: return "no exact device path";
  }
  return "unknown";
}

extern "C" const char *hg_last_error(const hg_ctx *ctx) {
  return ctx ? ctx->err.c_str() : g_create_err.c_str();
}

extern "C" const char *hg_version(void) { return "hypergen-hip 0.1.0 (gfx950)"; }

extern "C" int hg_device_count(void) {
  int n = 0;
  if (hipGetDeviceCount(&n) != hipSuccess) return 0;
  return n;
}

extern "C" hg_status hg_ctx_create(int device_id, hg_ctx **out) {
  if (!out) return hg_fail(nullptr, HG_ERR_INVALID, "out == NULL");
  *out = nullptr;
  int n = 0;
  hipError_t e = hipGetDeviceCount(&n);
  if (e != hipSuccess || n <= 0)
    return hg_fail(nullptr, HG_ERR_NO_DEVICE,
                   std::string("hipGetDeviceCount: ") + (e == hipSuccess ? "0 devices" : hipGetErrorString(e)));
  if (device_id < 0 || device_id >= n)
    return hg_fail(nullptr, HG_ERR_NO_DEVICE, "device id out of range");
  hg_ctx *c = new (std::nothrow) hg_ctx();
  if (!c) return hg_fail(nullptr, HG_ERR_OOM, "ctx allocation");
  c->device = device_id;
  if ((e = hipSetDevice(device_id)) != hipSuccess ||
      (e = hipStreamCreateWithFlags(&c->own_stream, hipStreamNonBlocking)) != hipSuccess) {
    delete c;
    return hg_fail(nullptr, HG_ERR_HIP, std::string("ctx setup: ") + hipGetErrorString(e));
  }
  c->stream = c->own_stream;
  hipDeviceProp_t prop;
  if (hipGetDeviceProperties(&prop, device_id) == hipSuccess) c->n_cu = prop.multiProcessorCount;
  *out = c;
  return HG_OK;
}

extern "C" void hg_ctx_destroy(hg_ctx *c) {
  if (!c) return;
  (void)hipSetDevice(c->device);
  (void)hipStreamSynchronize(c->stream);
  hg_ctx::Buf *bufs[] = {&c->w_items, &c->w_gmeta, &c->w_hits, &c->w_cnt, &c->w_seq, &c->w_hv, &c->w_hits2, &c->w_lsort, &c->w_redo, &c->w_pk, &c->w_pktab,
                         &c->w_misc, &c->w_f16a, &c->w_f16b, &c->w_stats, &c->w_ani, &c->w_hv2, &c->w_cen,
                         &c->w_n2a, &c->w_n2b, &c->w_sorthits, &c->w_i8a, &c->w_i8b, &c->w_i8misc};
  for (auto *b : bufs)
    if (b->p) (void)hipFree(b->p);
  for (auto &t : c->tile_tabs) {
    if (t.dev.p) (void)hipFree(t.dev.p);
    if (t.uploaded) (void)hipEventDestroy(t.uploaded);
  }
  for (auto &t : c->t_pending) {
    if (t.own_e0) (void)hipEventDestroy(t.e0);
    (void)hipEventDestroy(t.e1);
  }
  for (auto e : c->t_pool) (void)hipEventDestroy(e);
  if (c->h_res) (void)hipHostFree(c->h_res);
  if (c->h_pin) (void)hipHostFree(c->h_pin);
  if (c->h_chk) (void)hipHostFree(c->h_chk);
  if (c->h_plan) (void)hipHostFree(c->h_plan);
  if (c->plan_uploaded) (void)hipEventDestroy(c->plan_uploaded);
  for (auto e : c->copy_events) (void)hipEventDestroy(e);
  for (int i = 0; i < 2; ++i) {
    if (c->pack_buf[i]) (void)hipHostFree(c->pack_buf[i]);
    if (c->pack_ev[i]) (void)hipEventDestroy(c->pack_ev[i]);
  }
  if (c->copy_stream) (void)hipStreamDestroy(c->copy_stream);
  if (c->own_stream) (void)hipStreamDestroy(c->own_stream);
  delete c;
}

// The ctx workspaces (hit buffers, f16 operand copies, counters, the cached batch plan) are ordered by ONE stream.
// A call may return with its last kernels still running on them, so a switch to another stream first drains the
// old one: otherwise the next call's kernels could overwrite workspaces the old stream still reads.
static hg_status switch_stream(hg_ctx *c, hipStream_t to) {
  if (to == c->stream) return HG_OK;
  HG_ENTER(c);
  HG_HIP(c, hipStreamSynchronize(c->stream));
  c->stream = to;
  return HG_OK;
}

extern "C" hg_status hg_ctx_set_stream(hg_ctx *c, void *hip_stream) {
  if (!c) return HG_ERR_INVALID;
  return switch_stream(c, reinterpret_cast<hipStream_t>(hip_stream));  // NULL = HIP's default stream, on purpose
}

extern "C" hg_status hg_ctx_reset_stream(hg_ctx *c) {
  if (!c) return HG_ERR_INVALID;
  return switch_stream(c, c->own_stream);
}

extern "C" int hg_ctx_last_dist_path(const hg_ctx *c) { return c ? c->last_dist_path : -1; }

extern "C" hg_status hg_ctx_set_debug(hg_ctx *c, const char *key, const char *value) {
  if (!c || !key) return HG_ERR_INVALID;
  const std::string k = key, v = value ? value : "";
  if (k == "dist_tile") c->dbg_dist_tile = v;
  else if (k == "sort_test_buckets") c->dbg_sort_buckets = std::atoi(v.c_str());
  else if (k == "pair_limit") c->dbg_pair_limit = std::strtoull(v.c_str(), nullptr, 10);  // hg_dist_block_dev / hg_hamming_search_block_dev: row blocks from this many pairs on
  else if (k == "dist_path") c->dbg_dist_path = v;
  else if (k == "dist_order") c->dbg_dist_order = v;  // "plain": no diagonal-first tile order
  else if (k == "ham_path") c->dbg_ham_path = v;
  else if (k == "hostfed") c->dbg_hostfed = v;  // hg_sketch_batch / hg_kmer_hash_sample: "ascii" never 2-bit pack on the host, "packed" always
  else if (k == "sketch_path") c->dbg_sketch_path = v;  // "sync": every sketch step takes the synchronous path (counters read back between sort and encode)
  else if (k == "kmer_input") c->dbg_kmer_input = v;  // "packed": ASCII batches are 2-bit packed on the device first and take the packed kernels
  else return hg_fail(c, HG_ERR_INVALID, "unknown debug key " + k);
  return HG_OK;
}

extern "C" hg_status hg_ctx_sync(hg_ctx *c) {
  if (!c) return HG_ERR_INVALID;
  HG_ENTER(c);  // (a queued sketch step whose check word asks for it is run again here)
  HG_HIP(c, hipStreamSynchronize(c->stream));
  return HG_OK;
}

extern "C" hg_status hg_ctx_sketch_step_counts(hg_ctx *c, uint64_t *sync_free, uint64_t *synchronous, uint64_t *redone) {
  if (!c) return HG_ERR_INVALID;
  if (sync_free) *sync_free = c->n_fast_steps;
  if (synchronous) *synchronous = c->n_sync_steps;
  if (redone) *redone = c->n_redone_steps;
  return HG_OK;
}

static hipEvent_t take_event(hg_ctx *c) {
  if (!c->t_pool.empty()) {
    hipEvent_t e = c->t_pool.back();
    c->t_pool.pop_back();
    return e;
  }
  hipEvent_t e = nullptr;
  if (hipEventCreate(&e) != hipSuccess) return nullptr;
  return e;
}

hg_timed::hg_timed(hg_ctx *ctx, int cls_, int after_cls) : c(ctx), cls(cls_) {
  // (the chain is consumed by the next bracket whether or not that one records: a bracket opened while timing is off,
  // or one that never closes because its caller returned early, must not leave a stale event to chain to)
  hipEvent_t chain = c->t_chain;
  c->t_chain = nullptr;
  if (!c->timing) return;
  if (after_cls >= 0 && chain && c->t_chain_cls == after_cls) e0 = chain, own_e0 = false;
  else e0 = take_event(c);
  e1 = take_event(c);
  if (e0 && own_e0) (void)hipEventRecord(e0, c->stream);
}
hg_timed::~hg_timed() {
  if (!e0 || !e1) return;
  (void)hipEventRecord(e1, c->stream);
  c->t_pending.push_back({e0, e1, cls, own_e0});
  c->t_chain = e1, c->t_chain_cls = cls;
}

extern "C" hg_status hg_ctx_enable_timing(hg_ctx *c, int on) {
  if (!c) return HG_ERR_INVALID;
  c->timing = on != 0;
  c->t_chain = nullptr, c->t_chain_cls = -1;  // (nothing recorded before the switch may open a later bracket)
  return HG_OK;
}

extern "C" const char *hg_ctx_last_kernel(const hg_ctx *c, int cls) {
  return (c && cls >= 0 && cls < HG_T_COUNT) ? c->last_kernel[cls].c_str() : "";
}

extern "C" hg_status hg_ctx_timings(hg_ctx *c, float ms_sum[HG_T_COUNT], uint32_t launches[HG_T_COUNT]) {
  if (!c || !ms_sum || !launches) return HG_ERR_INVALID;
  for (int i = 0; i < HG_T_COUNT; ++i) ms_sum[i] = 0.f, launches[i] = 0;
  // read first, recycle afterwards -- on the error path too: every event goes back to the pool exactly once and the
  // pending list is emptied whatever happens (a second call must not pool the same events again)
  hipError_t first_err = hipSuccess;
  const char *first_what = "";
  for (auto &t : c->t_pending) {
    if (first_err != hipSuccess) break;
    hipError_t e = hipEventSynchronize(t.e1);
    float ms = 0.f;
    if (e == hipSuccess) e = hipEventElapsedTime(&ms, t.e0, t.e1);
    if (e != hipSuccess) {
      first_err = e, first_what = "hg_ctx_timings: event read";
      break;
    }
    if (t.cls >= 0 && t.cls < HG_T_COUNT) ms_sum[t.cls] += ms, launches[t.cls] += 1;
  }
  for (auto &t : c->t_pending) {  // (after the reads: a chained bracket reads its predecessor's e1)
    if (t.own_e0) c->t_pool.push_back(t.e0);
    c->t_pool.push_back(t.e1);
  }
  c->t_pending.clear();
  c->t_chain = nullptr;
  if (first_err != hipSuccess) return hg_fail(c, HG_ERR_HIP, std::string(first_what) + ": " + hipGetErrorString(first_err));
  return HG_OK;
}

hg_status hg_ensure(hg_ctx *c, hg_ctx::Buf &b, size_t bytes) {
  if (bytes <= b.cap) return HG_OK;
  HG_HIP(c, hipStreamSynchronize(c->stream));  // nothing in flight may still use the old block
  if (b.p) {
    (void)hipFree(b.p);
    b.p = nullptr;
    b.cap = 0;
  }
  size_t want = bytes + bytes / 4 + 256;
  hipError_t e = hipMalloc(&b.p, want);
  if (e != hipSuccess) {
    b.p = nullptr;
    return hg_fail(c, HG_ERR_OOM, "hipMalloc(" + std::to_string(want) + "): " + hipGetErrorString(e));
  }
  b.cap = want;
  return HG_OK;
}

namespace {
// zero_n > 0: the first zero_n device words (<= 64) are cleared behind the copy -- the call's counters are ready for the next
// call without a fill command of their own in the stream
__global__ void publish_words_kernel(uint32_t *__restrict__ src, uint32_t n, volatile uint32_t *dst, uint32_t seq, uint32_t zero_n) {
  if (threadIdx.x < n) dst[threadIdx.x] = src[threadIdx.x];
  if (threadIdx.x < zero_n) src[threadIdx.x] = 0u;
  __threadfence_system();
  __syncthreads();
  if (threadIdx.x == 0) {
    dst[16] = seq;
    __threadfence_system();
  }
}
}  // namespace

hg_status hg_publish_words(hg_ctx *c, uint32_t *d_words, uint32_t n, const uint32_t **out, uint32_t zero_n) {
  if (n > 16 || zero_n > 64) return hg_fail(c, HG_ERR_INVALID, "hg_publish_words: at most 16 words");
  if (!c->h_res) {
    void *p = nullptr;
    // (coherent = fine-grained: the device's writes and the fence between words and sequence number are seen by the polling host)
    hipError_t e = hipHostMalloc(&p, 32 * sizeof(uint32_t), hipHostMallocCoherent);
    if (e != hipSuccess) return hg_fail(c, HG_ERR_OOM, std::string("hipHostMalloc: ") + hipGetErrorString(e));
    c->h_res = static_cast<uint32_t *>(p);
    std::memset(c->h_res, 0, 32 * sizeof(uint32_t));
  }
  const uint32_t seq = ++c->res_seq ? c->res_seq : ++c->res_seq;  // never 0
  hipLaunchKernelGGL(publish_words_kernel, dim3(1), dim3(64), 0, c->stream, d_words, n, c->h_res, seq, zero_n);
  HG_HIP(c, hipGetLastError());
  volatile uint32_t *flag = c->h_res + 16;
  // Poll for a bounded time (the common case: a sub-millisecond GEMM, the word arrives within the poll), then hand the
  // core back: a search that runs for many milliseconds -- or one shard thread per GPU in hg_multi -- must not burn a host
  // core per ctx for its whole duration.
  const auto t0 = std::chrono::steady_clock::now();
  bool seen = false;
  for (uint32_t spins = 0; !seen; ++spins) {
    if (*flag == seq) {
      seen = true;
      break;
    }
#if defined(__x86_64__) || defined(__i386__)
    __builtin_ia32_pause();
#else
    std::this_thread::yield();
#endif
    if ((spins & 0x3ff) == 0x3ff &&
        std::chrono::steady_clock::now() - t0 > std::chrono::microseconds(800))
      break;
  }
  if (!seen) {
    const hipError_t e = hipStreamSynchronize(c->stream);  // blocks in the driver (interrupt), no spinning
    if (e != hipSuccess) return hg_fail(c, HG_ERR_HIP, std::string("hipStreamSynchronize: ") + hipGetErrorString(e));
    if (*flag != seq) return hg_fail(c, HG_ERR_HIP, "hg_publish_words: the stream finished without publishing");
  }
  std::atomic_thread_fence(std::memory_order_acquire);
  *out = c->h_res;
  return HG_OK;
}

hg_status hg_ensure_pinned(hg_ctx *c, size_t bytes) {
  if (bytes <= c->h_pin_cap) return HG_OK;
  HG_HIP(c, hipStreamSynchronize(c->stream));
  if (c->h_pin) (void)hipHostFree(c->h_pin);
  c->h_pin = nullptr, c->h_pin_cap = 0;
  size_t want = bytes + bytes / 4 + 4096;
  hipError_t e = hipHostMalloc(&c->h_pin, want, hipHostMallocDefault);
  if (e != hipSuccess) return hg_fail(c, HG_ERR_OOM, std::string("hipHostMalloc: ") + hipGetErrorString(e));
  c->h_pin_cap = want;
  return HG_OK;
}

extern "C" hg_status hg_dev_alloc(hg_ctx *c, size_t bytes, void **dptr) {
  if (!c || !dptr) return HG_ERR_INVALID;
  HG_ENTER(c);
  hipError_t e = hipMalloc(dptr, bytes ? bytes : 1);
  if (e != hipSuccess) return hg_fail(c, HG_ERR_OOM, std::string("hipMalloc: ") + hipGetErrorString(e));
  return HG_OK;
}
extern "C" hg_status hg_dev_free(hg_ctx *c, void *dptr) {
  if (!c) return HG_ERR_INVALID;
  HG_ENTER(c);  // (the block may be an operand of the queued sketch step)
  if (dptr) HG_HIP(c, hipFree(dptr));
  return HG_OK;
}
extern "C" hg_status hg_copy_h2d(hg_ctx *c, void *dst, const void *src, size_t bytes) {
  if (!c) return HG_ERR_INVALID;
  HG_ENTER(c);
  if (bytes) {
    HG_HIP(c, hipMemcpyAsync(dst, src, bytes, hipMemcpyHostToDevice, c->stream));
    HG_HIP(c, hipStreamSynchronize(c->stream));
  }
  return HG_OK;
}
extern "C" hg_status hg_copy_d2h(hg_ctx *c, void *dst, const void *src, size_t bytes) {
  if (!c) return HG_ERR_INVALID;
  HG_ENTER(c);
  if (bytes) {
    HG_HIP(c, hipMemcpyAsync(dst, src, bytes, hipMemcpyDeviceToHost, c->stream));
    HG_HIP(c, hipStreamSynchronize(c->stream));
  }
  return HG_OK;
}

extern "C" void hg_sketch_params_default(hg_sketch_params *p) {
  if (!p) return;
  std::memset(p, 0, sizeof *p);
  p->ksize = 21;  // src/types.rs:97-113
  p->canonical = 1;
  p->scaled = 1500;
  p->seed = 123;
  p->hv_d = 4096;
  p->hv_layout = HG_LAYOUT_AVX2;
  p->norm_mode = HG_NORM_ACGT;
}
