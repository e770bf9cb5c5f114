// hg_stream.hip -- continuous host-fed sketching (hg_sketch_stream_*).
//
// hg_sketch_batch overlaps upload and kernels INSIDE one call; a host that produces genomes one at a time (reader
// threads parsing FASTA files: src/sketch_cuda.rs:120-166 walks its file list the same way) pays the ends of every
// call -- the first upload with nothing to overlap, the last sub-batch's kernels and read-back with the link idle
// (~0.4 ms per call) -- and must collect a batch before it can call at all.  Here the two halves never stop:
//   * one uploader thread per device appends the pushed genomes to a ring of device chunks (~64 MB each) on a copy
//     stream; a chunk is handed on when it is full or the moment the input runs dry;
//   * one compute thread per device runs hg_sketch_batch_dev on the chunks as their uploads complete (stream
//     event) and queues the results.
// The PCIe link therefore carries sequence all the time, the kernels (a tenth of the upload time per genome) hide
// under it, and chunk sizes adapt by themselves: when the kernels lag, chunks fill up while the uploader waits.
#include <algorithm>
#include <chrono>
#include <condition_variable>
#include <cstring>
#include <deque>
#include <mutex>
#include <new>
#include <string>
#include <thread>
#include <vector>

#include "hg_internal.h"

namespace {

constexpr size_t CHUNK_BYTES = 64ull << 20;  // device bytes a chunk aims at (one larger genome still fits: the chunk grows)
constexpr size_t CHUNK_GENOMES = 4096;       // bounds the HV read-back of a chunk of tiny genomes (32 MiB at D = 4096)
constexpr size_t SMALL_BYTES = 256u << 10;   // genomes below this are packed into page-locked staging and uploaded together
constexpr int N_CHUNKS = 3;

struct Item {
  const uint8_t *seq;
  size_t len;
  uint64_t tag;
};

struct Chunk {
  uint8_t *d = nullptr;  // device sequence buffer
  size_t cap = 0;
  uint8_t *stage = nullptr;  // page-locked mirror for runs of small genomes (lazily allocated, CHUNK_BYTES)
  size_t run_lo = 0, run_hi = 0;
  std::vector<uint64_t> offs, lens, tags;
  size_t bytes = 0;
  hipEvent_t uploaded = nullptr;
};

struct Done {  // the results of one chunk
  std::vector<uint64_t> tags;
  std::vector<int16_t> hv;
  std::vector<int32_t> n2;
  std::vector<uint32_t> nh;
  size_t next = 0;
};

}  // namespace

struct hg_sketch_stream {
  struct Engine {
    int device = 0;
    hg_ctx *ctx = nullptr;
    hipStream_t copy = nullptr;
    Chunk chunk[N_CHUNKS];
    std::deque<Item> in;
    std::deque<int> free_chunks, full_chunks;
    size_t load = 0;  // bytes pushed to this engine whose results are not out yet
    bool uploader_done = false;
    int16_t *d_hv = nullptr;
    int32_t *d_n2 = nullptr;
    uint32_t *d_nh = nullptr;
    uint8_t *h_res = nullptr;  // page-locked read-back area
    std::thread up, comp;
    // diagnostics (hg_sketch_stream_stats): seconds spent per phase, chunks handed over
    double t_up_idle = 0, t_up_nochunk = 0, t_up_copy = 0, t_comp_idle = 0, t_comp_run = 0;
    size_t n_chunks = 0;
  };
  std::vector<Engine *> eng;
  hg_sketch_params p{};
  std::mutex mu;
  std::condition_variable cv_in, cv_chunk, cv_out, cv_room;
  std::deque<Done> out;
  size_t pushed = 0, popped = 0, max_pending = 4096;
  bool finishing = false;
  hg_status err = HG_OK;
  std::string msg;
};

namespace {

using Engine = hg_sketch_stream::Engine;

double now_s() { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); }

void fail(hg_sketch_stream *s, hg_status st, const std::string &m) {
  std::lock_guard<std::mutex> lk(s->mu);
  if (s->err == HG_OK) s->err = st, s->msg = m;
  s->cv_in.notify_all(), s->cv_chunk.notify_all(), s->cv_out.notify_all(), s->cv_room.notify_all();
}

#define ST_HIP(s, expr)                                                                    \
  do {                                                                                     \
    hipError_t e__ = (expr);                                                               \
    if (e__ != hipSuccess) {                                                               \
      fail((s), HG_ERR_HIP, std::string(#expr) + ": " + hipGetErrorString(e__));           \
      return false;                                                                        \
    }                                                                                      \
  } while (0)

// the pending run of small genomes of chunk c goes up as one copy
bool flush_run(hg_sketch_stream *s, Engine &e, Chunk &c) {
  if (c.run_hi > c.run_lo)
    ST_HIP(s, hipMemcpyAsync(c.d + c.run_lo, c.stage + c.run_lo, c.run_hi - c.run_lo, hipMemcpyHostToDevice, e.copy));
  c.run_lo = c.run_hi = 0;
  return true;
}

bool hand_over(hg_sketch_stream *s, Engine &e, int ci) {
  Chunk &c = e.chunk[ci];
  if (!flush_run(s, e, c)) return false;
  ST_HIP(s, hipEventRecord(c.uploaded, e.copy));
  std::lock_guard<std::mutex> lk(s->mu);
  e.full_chunks.push_back(ci);
  s->cv_chunk.notify_all();
  return true;
}

void uploader(hg_sketch_stream *s, Engine *ep) {
  Engine &e = *ep;
  auto body = [&]() -> bool {
    ST_HIP(s, hipSetDevice(e.device));
    int cur = -1;
    for (;;) {
      Item it{};
      bool idle_after;
      {
        const double tw = now_s();
        std::unique_lock<std::mutex> lk(s->mu);
        s->cv_in.wait(lk, [&] { return !e.in.empty() || s->finishing || s->err != HG_OK; });
        e.t_up_idle += now_s() - tw;
        if (s->err != HG_OK) return false;
        if (e.in.empty()) break;  // finishing
        it = e.in.front();
        e.in.pop_front();
        idle_after = e.in.empty();
      }
      const size_t padded = (it.len + 15) & ~(size_t)15;
      if (cur >= 0 && e.chunk[cur].bytes &&
          (e.chunk[cur].bytes + padded > CHUNK_BYTES || e.chunk[cur].tags.size() >= CHUNK_GENOMES)) {
        if (!hand_over(s, e, cur)) return false;
        cur = -1;
      }
      if (cur < 0) {
        const double tw = now_s();
        std::unique_lock<std::mutex> lk(s->mu);
        s->cv_chunk.wait(lk, [&] { return !e.free_chunks.empty() || s->err != HG_OK; });
        e.t_up_nochunk += now_s() - tw;
        if (s->err != HG_OK) return false;
        cur = e.free_chunks.front();
        e.free_chunks.pop_front();
        Chunk &c = e.chunk[cur];
        c.offs.clear(), c.lens.clear(), c.tags.clear();
        c.bytes = 0, c.run_lo = c.run_hi = 0;
      }
      Chunk &c = e.chunk[cur];
      if (c.bytes + padded + 64 > c.cap) {  // one genome larger than the chunk (bytes == 0 here): the chunk grows
        if (c.d) ST_HIP(s, hipFree(c.d));
        c.d = nullptr, c.cap = 0;
        const size_t want = padded + padded / 8 + 64;
        hipError_t he = hipMalloc(reinterpret_cast<void **>(&c.d), want);
        if (he != hipSuccess) {
          fail(s, HG_ERR_OOM, "hipMalloc(" + std::to_string(want) + "): " + hipGetErrorString(he));
          return false;
        }
        c.cap = want;
      }
      const double tc = now_s();
      if (it.len) {
        if (it.len < SMALL_BYTES && c.bytes + padded <= CHUNK_BYTES) {
          if (!c.stage) ST_HIP(s, hipHostMalloc(reinterpret_cast<void **>(&c.stage), CHUNK_BYTES, hipHostMallocDefault));
          if (c.run_hi == c.run_lo) c.run_lo = c.run_hi = c.bytes;
          std::memcpy(c.stage + c.bytes, it.seq, it.len);
          if (padded > it.len) std::memset(c.stage + c.bytes + it.len, 0, padded - it.len);
          c.run_hi = c.bytes + padded;
        } else {
          if (!flush_run(s, e, c)) return false;
          ST_HIP(s, hipMemcpyAsync(c.d + c.bytes, it.seq, it.len, hipMemcpyHostToDevice, e.copy));
        }
      }
      e.t_up_copy += now_s() - tc;
      c.offs.push_back(c.bytes), c.lens.push_back(it.len), c.tags.push_back(it.tag);
      c.bytes += padded;
      // hand the chunk on when it is full -- or when nothing else is waiting: the kernels start at once and the
      // next genome opens a new chunk
      if (idle_after || c.bytes >= CHUNK_BYTES || c.tags.size() >= CHUNK_GENOMES) {
        if (!hand_over(s, e, cur)) return false;
        cur = -1;
      }
    }
    if (cur >= 0 && !e.chunk[cur].tags.empty() && !hand_over(s, e, cur)) return false;
    return true;
  };
  (void)body();
  std::lock_guard<std::mutex> lk(s->mu);
  e.uploader_done = true;
  s->cv_chunk.notify_all();
}

void computer(hg_sketch_stream *s, Engine *ep) {
  Engine &e = *ep;
  auto body = [&]() -> bool {
    ST_HIP(s, hipSetDevice(e.device));
    const size_t D = s->p.hv_d;
    for (;;) {
      int ci;
      const double tw = now_s();
      {
        std::unique_lock<std::mutex> lk(s->mu);
        s->cv_chunk.wait(lk, [&] { return !e.full_chunks.empty() || e.uploader_done || s->err != HG_OK; });
        e.t_comp_idle += now_s() - tw;
        if (s->err != HG_OK) return false;
        if (e.full_chunks.empty()) break;  // the uploader is done and so are we
        ci = e.full_chunks.front();
        e.full_chunks.pop_front();
      }
      Chunk &c = e.chunk[ci];
      const size_t m = c.tags.size();
      const double tr = now_s();
      ST_HIP(s, hipStreamWaitEvent(e.ctx->stream, c.uploaded, 0));
      const hg_status st = hg_sketch_batch_dev(e.ctx, c.d, c.offs.data(), c.lens.data(), m, &s->p, e.d_hv, e.d_n2, e.d_nh);
      if (st != HG_OK) {
        fail(s, st, std::string("device ") + std::to_string(e.device) + ": " + hg_last_error(e.ctx));
        return false;
      }
      const size_t hvb = m * D * sizeof(int16_t), hvb_al = (CHUNK_GENOMES * D * sizeof(int16_t) + 63) & ~(size_t)63;
      ST_HIP(s, hipMemcpyAsync(e.h_res, e.d_hv, hvb, hipMemcpyDeviceToHost, e.ctx->stream));
      ST_HIP(s, hipMemcpyAsync(e.h_res + hvb_al, e.d_n2, m * 4, hipMemcpyDeviceToHost, e.ctx->stream));
      ST_HIP(s, hipMemcpyAsync(e.h_res + hvb_al + CHUNK_GENOMES * 4, e.d_nh, m * 4, hipMemcpyDeviceToHost, e.ctx->stream));
      ST_HIP(s, hipStreamSynchronize(e.ctx->stream));
      Done d;
      d.tags = c.tags;
      d.hv.assign(reinterpret_cast<int16_t *>(e.h_res), reinterpret_cast<int16_t *>(e.h_res) + m * D);
      d.n2.assign(reinterpret_cast<int32_t *>(e.h_res + hvb_al), reinterpret_cast<int32_t *>(e.h_res + hvb_al) + m);
      d.nh.assign(reinterpret_cast<uint32_t *>(e.h_res + hvb_al + CHUNK_GENOMES * 4),
                  reinterpret_cast<uint32_t *>(e.h_res + hvb_al + CHUNK_GENOMES * 4) + m);
      std::lock_guard<std::mutex> lk(s->mu);
      e.t_comp_run += now_s() - tr, ++e.n_chunks;
      e.load -= std::min(e.load, c.bytes);
      s->out.push_back(std::move(d));
      e.free_chunks.push_back(ci);
      s->cv_chunk.notify_all(), s->cv_out.notify_all();
    }
    return true;
  };
  (void)body();
  std::lock_guard<std::mutex> lk(s->mu);
  s->cv_out.notify_all();
}

void destroy(hg_sketch_stream *s) {
  {
    std::lock_guard<std::mutex> lk(s->mu);
    s->finishing = true;
    if (s->err == HG_OK && s->popped < s->pushed) s->err = HG_ERR_INVALID, s->msg = "stream closed with results outstanding";
    s->cv_in.notify_all(), s->cv_chunk.notify_all(), s->cv_out.notify_all(), s->cv_room.notify_all();
  }
  for (Engine *e : s->eng) {
    if (e->up.joinable()) e->up.join();
    if (e->comp.joinable()) e->comp.join();
    (void)hipSetDevice(e->device);
    if (e->copy) (void)hipStreamSynchronize(e->copy);
    for (Chunk &c : e->chunk) {
      if (c.d) (void)hipFree(c.d);
      if (c.stage) (void)hipHostFree(c.stage);
      if (c.uploaded) (void)hipEventDestroy(c.uploaded);
    }
    if (e->d_hv) (void)hipFree(e->d_hv);
    if (e->h_res) (void)hipHostFree(e->h_res);
    if (e->copy) (void)hipStreamDestroy(e->copy);
    if (e->ctx) hg_ctx_destroy(e->ctx);
    delete e;
  }
  delete s;
}

}  // namespace

extern "C" hg_status hg_sketch_stream_open(const int *device_ids, int n_devices, const hg_sketch_params *p,
                                           hg_sketch_stream **out) {
  if (!out) return HG_ERR_INVALID;
  *out = nullptr;
  if (!device_ids || n_devices <= 0 || !p) return hg_fail(nullptr, HG_ERR_INVALID, "hg_sketch_stream_open: bad arguments");
  if (p->hv_d == 0 || p->hv_d > 32768) return hg_fail(nullptr, HG_ERR_UNSUPPORTED, "hv_d must be in 1..32768");
  hg_sketch_stream *s = new (std::nothrow) hg_sketch_stream();
  if (!s) return HG_ERR_OOM;
  s->p = *p;
  const size_t D = p->hv_d;
  const size_t hvb_al = (CHUNK_GENOMES * D * sizeof(int16_t) + 63) & ~(size_t)63;
  for (int i = 0; i < n_devices; ++i) {
    Engine *e = new (std::nothrow) Engine();
    if (!e) {
      destroy(s);
      return HG_ERR_OOM;
    }
    s->eng.push_back(e);
    e->device = device_ids[i];
    hg_status st = hg_ctx_create(device_ids[i], &e->ctx);
    hipError_t he = hipSuccess;
    if (st == HG_OK) {
      if ((he = hipSetDevice(e->device)) == hipSuccess)
        he = hipStreamCreateWithFlags(&e->copy, hipStreamNonBlocking);
      for (int k = 0; k < N_CHUNKS && he == hipSuccess; ++k) {
        if ((he = hipMalloc(reinterpret_cast<void **>(&e->chunk[k].d), CHUNK_BYTES + 64)) == hipSuccess)
          e->chunk[k].cap = CHUNK_BYTES + 64, he = hipEventCreateWithFlags(&e->chunk[k].uploaded, hipEventDisableTiming);
        e->free_chunks.push_back(k);
      }
      void *dres = nullptr;
      if (he == hipSuccess) he = hipMalloc(&dres, hvb_al + CHUNK_GENOMES * 8);
      if (he == hipSuccess) {
        e->d_hv = static_cast<int16_t *>(dres);
        e->d_n2 = reinterpret_cast<int32_t *>(static_cast<uint8_t *>(dres) + hvb_al);
        e->d_nh = reinterpret_cast<uint32_t *>(e->d_n2 + CHUNK_GENOMES);
        he = hipHostMalloc(reinterpret_cast<void **>(&e->h_res), hvb_al + CHUNK_GENOMES * 8, hipHostMallocDefault);
      }
    }
    if (st != HG_OK || he != hipSuccess) {
      const std::string m = st != HG_OK ? std::string(hg_last_error(nullptr)) : std::string("stream setup: ") + hipGetErrorString(he);
      destroy(s);
      return hg_fail(nullptr, st != HG_OK ? st : HG_ERR_HIP, m);
    }
  }
  for (Engine *e : s->eng) {
    e->up = std::thread(uploader, s, e);
    e->comp = std::thread(computer, s, e);
  }
  *out = s;
  return HG_OK;
}

extern "C" hg_status hg_sketch_stream_push(hg_sketch_stream *s, const uint8_t *seq, size_t len, uint64_t tag) {
  if (!s || (len && !seq)) return HG_ERR_INVALID;
  std::unique_lock<std::mutex> lk(s->mu);
  if (s->finishing) return HG_ERR_INVALID;
  s->cv_room.wait(lk, [&] { return s->pushed - s->popped < s->max_pending || s->err != HG_OK; });
  if (s->err != HG_OK) return s->err;
  Engine *best = s->eng[0];
  for (Engine *e : s->eng)
    if (e->load < best->load) best = e;
  best->in.push_back(Item{seq, len, tag});
  best->load += (len + 15) & ~(size_t)15;
  ++s->pushed;
  s->cv_in.notify_all();
  return HG_OK;
}

extern "C" hg_status hg_sketch_stream_finish(hg_sketch_stream *s) {
  if (!s) return HG_ERR_INVALID;
  std::lock_guard<std::mutex> lk(s->mu);
  s->finishing = true;
  s->cv_in.notify_all(), s->cv_out.notify_all();
  return s->err;
}

extern "C" hg_status hg_sketch_stream_pop(hg_sketch_stream *s, uint64_t *tag, int16_t *hv_out, int32_t *norm2_out,
                                          uint32_t *nhash_out, int *got) {
  if (!s || !got) return HG_ERR_INVALID;
  *got = 0;
  std::unique_lock<std::mutex> lk(s->mu);
  s->cv_out.wait(lk, [&] { return !s->out.empty() || s->err != HG_OK || (s->finishing && s->popped == s->pushed); });
  if (s->out.empty()) return s->err;  // failed, or finished and drained (HG_OK, *got == 0)
  Done &d = s->out.front();
  const size_t k = d.next++, D = s->p.hv_d;
  if (tag) *tag = d.tags[k];
  if (hv_out) std::memcpy(hv_out, d.hv.data() + k * D, D * sizeof(int16_t));
  if (norm2_out) *norm2_out = d.n2[k];
  if (nhash_out) *nhash_out = d.nh[k];
  if (d.next == d.tags.size()) s->out.pop_front();
  ++s->popped;
  *got = 1;
  s->cv_room.notify_all();
  if (s->finishing && s->popped == s->pushed) s->cv_out.notify_all();
  return HG_OK;
}

extern "C" const char *hg_sketch_stream_last_error(hg_sketch_stream *s) {
  if (!s) return "";
  std::lock_guard<std::mutex> lk(s->mu);
  return s->msg.c_str();
}

extern "C" hg_status hg_sketch_stream_stats(hg_sketch_stream *s, int engine, double out[6]) {
  if (!s || !out || engine < 0 || engine >= (int)s->eng.size()) return HG_ERR_INVALID;
  std::lock_guard<std::mutex> lk(s->mu);
  const Engine &e = *s->eng[engine];
  out[0] = e.t_up_idle, out[1] = e.t_up_nochunk, out[2] = e.t_up_copy, out[3] = e.t_comp_idle, out[4] = e.t_comp_run;
  out[5] = (double)e.n_chunks;
  return HG_OK;
}

extern "C" void hg_sketch_stream_close(hg_sketch_stream *s) {
  if (s) destroy(s);
}
