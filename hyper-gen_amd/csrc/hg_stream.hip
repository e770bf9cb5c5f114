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
#include <cstdlib>
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
// compute threads (each with its own ctx) per device.  A/B at 2 (+ a 4th chunk): 7.3-7.8 k files/s instead of
// 8.3-9.6 k -- two contexts' small kernels and synchronisations get in each other's way; kept configurable
constexpr int N_WORKERS = 1;

enum : int { KIND_ASCII = 0, KIND_PACK2 = 1, KIND_PACK2S = 2 };
struct Item {
  const uint8_t *seq;  // ASCII sequence, a hg_pack2 blob (KIND_PACK2) or a hg_pack2s blob (KIND_PACK2S)
  size_t len;          // bases
  uint64_t tag;
  int kind;
  size_t blob_bytes;   // KIND_PACK2S: bytes of the host blob (codes + run table)
};

// one packed genome of a chunk: where its blob sits in the chunk's packed area, where its ASCII goes
struct UnpackJob {
  uint64_t pk_off, out_off, n_bps, mask_off;  // (mask_off: where the genome's not-a-base bitmap lies in the packed area)
  uint32_t first_block, pad;
};
// one sparse genome of a chunk: its run table lies behind its codes, its bitmap is rebuilt at mask_off
struct SparseJob {
  uint64_t codes_off, mask_off, n_bps;
  uint32_t first_block, pad;
};
constexpr uint32_t SLICE_WORDS = 1024;  // bitmap words one workgroup rebuilds: 4 KiB = 32 768 bases
constexpr uint32_t UNPACK_GROUPS_PER_BLOCK = 1024;  // 256 threads x 4 groups of 16 bases

struct Chunk {
  uint8_t *d = nullptr;  // device sequence buffer
  size_t cap = 0;
  uint8_t *stage = nullptr;  // page-locked mirror for runs of small genomes (lazily allocated, CHUNK_BYTES)
  size_t run_lo = 0, run_hi = 0;
  std::vector<uint64_t> offs, lens, tags;
  std::vector<uint64_t> pk_offs;  // per genome: its blob's offset in dpk (packed genomes only; parallel to offs when ALL are packed)
  size_t bytes = 0;
  hipEvent_t uploaded = nullptr;
  // hg_pack2 blobs (hg_sketch_stream_push_packed): their own device area.  A chunk made of blobs only goes to the
  // packed-input kernels as it is (hg_sketch_batch_dev_packed); a chunk that mixes blobs and ASCII genomes has its blobs
  // expanded into `d` by unpack2_kernel first
  uint8_t *dpk = nullptr;
  size_t pk_cap = 0, pk_bytes = 0;
  UnpackJob *h_jobs = nullptr, *d_jobs = nullptr;  // CHUNK_GENOMES entries each (page-locked / device), lazily allocated
  uint32_t n_jobs = 0, n_blocks = 0;
  SparseJob *h_sjobs = nullptr, *d_sjobs = nullptr;  // hg_pack2s genomes: bitmap rebuild jobs
  uint32_t n_sjobs = 0, n_sblocks = 0;
  std::vector<uint64_t> mask_offs;  // parallel to pk_offs
  bool has_ascii = false;           // some genome of the chunk arrived as ASCII (its bytes live in `d`)
  size_t link_bytes = 0;            // bytes this chunk moves over the link
};

struct Done {  // the results of one chunk
  std::vector<uint64_t> tags;
  std::vector<int16_t> hv;
  std::vector<int32_t> n2;
  std::vector<uint32_t> nh;
  size_t next = 0;
};

}  // namespace

namespace {
// 16 bases per step: 4 code bytes -> 16 ASCII bytes through a v_perm table ("ACGT"), non-bases -> 'N'
__device__ __forceinline__ void unpack2_group(const uint8_t *__restrict__ blob, const uint8_t *__restrict__ mask, uint8_t *__restrict__ out,
                                              uint64_t grp) {
  const uint32_t codes = *reinterpret_cast<const uint32_t *>(blob + 4 * grp);
  const uint32_t bad = *reinterpret_cast<const uint16_t *>(mask + 2 * grp);
  uint32_t w[4];
#pragma unroll
  for (int b = 0; b < 4; ++b) {
    const uint32_t c = (codes >> (8 * b)) & 0xFFu;
    const uint32_t sel = (c & 3u) | ((c & 0xCu) << 6) | ((c & 0x30u) << 12) | ((c & 0xC0u) << 18);  // 2-bit fields -> bytes
    const uint32_t ascii = __builtin_amdgcn_perm(0u, 0x54474341u, sel);  // selector 0..3 -> 'A','C','G','T'
    const uint32_t m = ((((bad >> (4 * b)) & 0xFu) * 0x00204081u) & 0x01010101u) * 0xFFu;  // mask bits -> byte masks
    w[b] = (ascii & ~m) | (0x4E4E4E4Eu & m);
  }
  *reinterpret_cast<uint4 *>(out + 16 * grp) = make_uint4(w[0], w[1], w[2], w[3]);
}

__global__ __launch_bounds__(256) void unpack2_kernel(const uint8_t *__restrict__ pk, uint8_t *__restrict__ out,
                                                      const UnpackJob *__restrict__ jobs, uint32_t n_jobs) {
  uint32_t lo = 0, hi = n_jobs;  // the last job whose first block is <= blockIdx.x
  while (hi - lo > 1) {
    const uint32_t mid = (lo + hi) >> 1;
    if (jobs[mid].first_block <= blockIdx.x) lo = mid;
    else hi = mid;
  }
  const UnpackJob jb = jobs[lo];
  const uint64_t groups = (jb.n_bps + 15) / 16;
  const uint64_t g0 = (uint64_t)(blockIdx.x - jb.first_block) * UNPACK_GROUPS_PER_BLOCK + threadIdx.x;
#pragma unroll
  for (int r = 0; r < 4; ++r) {
    const uint64_t g = g0 + 256u * r;
    if (g < groups) unpack2_group(pk + jb.pk_off, pk + jb.mask_off, out + jb.out_off, g);
  }
}

// hg_pack2s genomes: the not-a-base bitmap of the hg_pack2 layout rebuilt from the run table that came over the link.
// One workgroup per 4 KiB slice of a genome's bitmap: zeroed in LDS, the runs that overlap it OR-ed in, written out
// once -- the chunk's memory is reused, so every word is written whether it has a bit or not.
__global__ __launch_bounds__(256) void expand_runs_kernel(uint8_t *__restrict__ pk, const SparseJob *__restrict__ jobs, uint32_t n_jobs) {
  __shared__ uint32_t s_bits[SLICE_WORDS];
  uint32_t lo = 0, hi = n_jobs;
  while (hi - lo > 1) {
    const uint32_t mid = (lo + hi) >> 1;
    if (jobs[mid].first_block <= blockIdx.x) lo = mid;
    else hi = mid;
  }
  const SparseJob jb = jobs[lo];
  const uint64_t cb = ((jb.n_bps + 3) / 4 + 15) & ~(uint64_t)15, words = (((jb.n_bps + 7) / 8 + 15) & ~(uint64_t)15) / 4;
  const uint64_t w0 = (uint64_t)(blockIdx.x - jb.first_block) * SLICE_WORDS;
  if (w0 >= words) return;
  const uint32_t nw = (uint32_t)(words - w0 < SLICE_WORDS ? words - w0 : SLICE_WORDS);
  for (uint32_t i = threadIdx.x; i < nw; i += 256) s_bits[i] = 0u;
  __syncthreads();
  const uint32_t *__restrict__ tab = reinterpret_cast<const uint32_t *>(pk + jb.codes_off + cb);
  const uint32_t n_runs = tab[0];
  const uint64_t b0 = 32 * w0, b1 = b0 + 32ull * nw;
  uint32_t a = 0, z = n_runs;  // first run that ends behind b0 (runs are sorted and disjoint)
  while (a < z) {
    const uint32_t mid = (a + z) >> 1;
    if ((uint64_t)tab[2 + 2 * mid] + tab[3 + 2 * mid] > b0) z = mid;
    else a = mid + 1;
  }
  for (uint32_t r = a; r < n_runs; ++r) {  // uniform: a slice sees a handful of runs
    const uint64_t st = tab[2 + 2 * r], en = st + tab[3 + 2 * r];
    if (st >= b1) break;
    const uint64_t s_ = st > b0 ? st : b0, e_ = en < b1 ? en : b1;  // e_ > s_
    const uint32_t fw = (uint32_t)((s_ - b0) >> 5), lw = (uint32_t)((e_ - 1 - b0) >> 5);
    for (uint32_t w = fw + threadIdx.x; w <= lw; w += 256) {
      uint32_t m = ~0u;
      if (w == fw) m &= ~0u << (uint32_t)(s_ & 31);
      if (w == lw) m &= ~0u >> (31u - (uint32_t)((e_ - 1) & 31));
      atomicOr(&s_bits[w], m);
    }
  }
  __syncthreads();
  uint32_t *__restrict__ dst = reinterpret_cast<uint32_t *>(pk + jb.mask_off) + w0;
  for (uint32_t i = threadIdx.x; i < nw; i += 256) dst[i] = s_bits[i];
}

// ASCII -> hg_pack2 blob, 32 bases per lane (8 code bytes + 4 bitmap bytes), bit-identical to the host's hg_pack2
// (hg_formats.cpp): A,C,G,T = 0..3 in either case (+ u/U -> T under u2t), anything else code 0 + its not-a-base bit;
// the paddings of both areas (to 16 bytes) and everything behind the last base are zero.  tab: {seq_off, n_bps, blob_off}
// per genome; blockIdx.y = genome.
__global__ __launch_bounds__(256) void pack2_kernel(const uint8_t *__restrict__ seq, const uint64_t *__restrict__ tab,
                                                    uint32_t u2t, uint8_t *__restrict__ blobs) {
  const uint64_t seq_off = tab[3 * blockIdx.y], n = tab[3 * blockIdx.y + 1], blob_off = tab[3 * blockIdx.y + 2];
  const uint64_t cb = ((n + 3) / 4 + 15) & ~(uint64_t)15, mb = ((n + 7) / 8 + 15) & ~(uint64_t)15;
  const uint64_t q = (uint64_t)blockIdx.x * 256 + threadIdx.x, i0 = 32 * q;
  if (4 * q >= mb) return;  // (the bitmap's padding reaches further than the codes')
  const uint8_t *__restrict__ src = seq + seq_off;
  uint32_t x[8];
  if (i0 + 32 <= n) {
    const uint32_t *s4 = reinterpret_cast<const uint32_t *>(src + i0);  // seq_off is a multiple of 4
#pragma unroll
    for (int t = 0; t < 8; ++t) x[t] = s4[t];
  } else {
#pragma unroll
    for (int t = 0; t < 8; ++t) {
      uint32_t w = 0;
      for (int b = 0; b < 4; ++b) {
        const uint64_t i = i0 + 4 * t + b;
        w |= (uint32_t)(i < n ? src[i] : (uint8_t)0) << (8 * b);
      }
      x[t] = w;
    }
  }
  uint32_t codes[2] = {0u, 0u}, bad = 0u;
#pragma unroll
  for (int t = 0; t < 8; ++t) {
    uint32_t xv = x[t];
    if (u2t) {  // u/U -> T ('U' ^ 'T' == 1)
      const uint32_t e = (xv & 0xDFDFDFDFu) ^ 0x55555555u;
      const uint32_t nz = ((e & 0x7F7F7F7Fu) + 0x7F7F7F7Fu) | e;  // bit 7 set <=> byte != 'U'
      xv ^= (~nz & 0x80808080u) >> 7;
    }
    const uint32_t tt = xv ^ (xv >> 1);
    uint32_t cd = (tt >> 1) & 0x03030303u;
    const uint32_t d = (xv & 0xDFDFDFDFu) ^ __builtin_amdgcn_perm(0u, 0x54474341u, cd);
    const uint32_t z = ((((d & 0x7F7F7F7Fu) + 0x7F7F7F7Fu) | d) & 0x80808080u) >> 7;  // 1 per byte that is not a base
    cd &= ~(z * 0xFFu);
    codes[t >> 2] |= ((cd | (cd >> 6) | (cd >> 12) | (cd >> 18)) & 0xFFu) << (8 * (t & 3));
    uint32_t nb = ((z * 0x01020408u) >> 24) & 0xFu;
    // positions at or behind the end are not flagged (the host leaves those bits zero)
    const uint64_t p0 = i0 + 4 * t;
    if (p0 + 4 > n) nb &= p0 >= n ? 0u : ((1u << (uint32_t)(n - p0)) - 1u);
    bad |= nb << (4 * t);
  }
  uint8_t *blob = blobs + blob_off;
  if (8 * q < cb) *reinterpret_cast<uint2 *>(blob + 8 * q) = make_uint2(codes[0], codes[1]);
  *reinterpret_cast<uint32_t *>(blob + cb + 4 * q) = bad;
}

__global__ __launch_bounds__(256) void unpack2_one_kernel(const uint8_t *__restrict__ blob, uint8_t *__restrict__ out, uint64_t n_bps) {
  const uint64_t groups = (n_bps + 15) / 16;
  const size_t code_bytes = (((size_t)n_bps + 3) / 4 + 15) & ~(size_t)15;
  const uint64_t g0 = (uint64_t)blockIdx.x * UNPACK_GROUPS_PER_BLOCK + threadIdx.x;
#pragma unroll
  for (int r = 0; r < 4; ++r) {
    const uint64_t g = g0 + 256u * r;
    if (g < groups) unpack2_group(blob, blob + code_bytes, out, g);
  }
}
}  // namespace

struct hg_sketch_stream {
  struct Engine {
    int device = 0;
    struct Worker {  // one compute thread: its own ctx (workspaces, stream) and result areas
      hg_ctx *ctx = nullptr;
      int16_t *d_hv = nullptr;
      int32_t *d_n2 = nullptr;
      uint32_t *d_nh = nullptr;
      uint8_t *h_res = nullptr;  // page-locked read-back area
      std::thread th;
    } w[N_WORKERS];
    hipStream_t copy = nullptr;
    // A second copy stream for the genomes' own uploads, used in turn with `copy`: a copy command costs ~7 us of link
    // idle time whatever it moves (ASCII 5 MB: 50 GB/s; hg_pack2 1.9 MB: 42; hg_pack2s 1.25 MB: 36 -- the bench's
    // packed_stream legs), and with two queues the command overhead of one runs under the transfer of the other.
    hipStream_t copy2 = nullptr;
    hipEvent_t copy2_done = nullptr;  // "everything queued on copy2 for the chunk being handed over"
    unsigned turn = 0;
    Chunk chunk[N_CHUNKS];
    std::deque<Item> in;
    std::deque<int> free_chunks, full_chunks;
    size_t load = 0;  // bytes pushed to this engine whose results are not out yet
    bool uploader_done = false;
    std::thread up;
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
  if (c.n_jobs) ST_HIP(s, hipMemcpyAsync(c.d_jobs, c.h_jobs, c.n_jobs * sizeof(UnpackJob), hipMemcpyHostToDevice, e.copy));
  if (c.n_sjobs) ST_HIP(s, hipMemcpyAsync(c.d_sjobs, c.h_sjobs, c.n_sjobs * sizeof(SparseJob), hipMemcpyHostToDevice, e.copy));
  ST_HIP(s, hipEventRecord(e.copy2_done, e.copy2));  // the chunk's uploads on the second stream join the first
  ST_HIP(s, hipStreamWaitEvent(e.copy, e.copy2_done, 0));
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
    // Chunk size ramp: after the input ran dry the first chunk of a new burst closes at CHUNK_BYTES / 8 and every
    // further one at twice the previous size (up to CHUNK_BYTES), so that the kernels start ~0.15 ms after the burst
    // does instead of after a whole 64 MB upload -- with a burst of a few hundred genomes that idle start was a fifth
    // of the pass (256 genomes x 1.25 MB: 28.7 k genomes/s; the bench's packed_stream leg)
    size_t limit = CHUNK_BYTES / 8;
    for (;;) {
      Item it{};
      bool idle_after;
      {
        const double tw = now_s();
        std::unique_lock<std::mutex> lk(s->mu);
        if (e.in.empty() && cur < 0) limit = CHUNK_BYTES / 8;  // (about to wait with nothing open: a new burst)
        s->cv_in.wait(lk, [&] { return !e.in.empty() || s->finishing || s->err != HG_OK; });
        e.t_up_idle += now_s() - tw;
        if (s->err != HG_OK) return false;
        if (e.in.empty()) break;  // finishing
        it = e.in.front();
        e.in.pop_front();
        idle_after = e.in.empty();
      }
      const size_t padded = (it.len + 15) & ~(size_t)15;
      // what the genome moves over the link (and what a chunk is sized by): its ASCII bytes or its blob
      const size_t link = it.kind == KIND_ASCII ? padded : (it.kind == KIND_PACK2 ? hg_pack2_size(it.len) : it.blob_bytes);
      if (cur >= 0 && !e.chunk[cur].tags.empty()) {
        const Chunk &cc = e.chunk[cur];
        // a chunk that holds (or is about to hold) ASCII genomes is bounded by its ASCII buffer `d`; a chunk of blobs only
        // never touches `d` and is bounded by the bytes it uploads -- three to four times as many genomes per chunk, so
        // that the fixed cost of a chunk (two stream synchronisations, ~0.3 ms) is shared by more of them
        const bool ascii_rule = cc.has_ascii || it.kind == KIND_ASCII;
        const bool full = ascii_rule ? cc.bytes + padded > limit : cc.link_bytes + link > limit;
        if (full || cc.tags.size() >= CHUNK_GENOMES) {
          if (!hand_over(s, e, cur)) return false;
          cur = -1, limit = std::min(CHUNK_BYTES, 2 * limit);
        }
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
        c.offs.clear(), c.lens.clear(), c.tags.clear(), c.pk_offs.clear();
        c.bytes = 0, c.run_lo = c.run_hi = 0;
        c.pk_bytes = 0, c.n_jobs = 0, c.n_blocks = 0, c.n_sjobs = 0, c.n_sblocks = 0;
        c.mask_offs.clear(), c.has_ascii = false, c.link_bytes = 0;
      }
      Chunk &c = e.chunk[cur];
      if (it.kind == KIND_ASCII) c.has_ascii = true;
      if ((it.kind == KIND_ASCII || c.has_ascii) && c.bytes + padded + 64 > c.cap) {  // one genome larger than the chunk (bytes == 0 here): the chunk grows
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
      if (it.len && it.kind != KIND_ASCII) {
        const size_t cbytes = (((it.len + 3) / 4) + 15) & ~(size_t)15, mbytes = (((it.len + 7) / 8) + 15) & ~(size_t)15;
        // device region: hg_pack2 = [codes][bitmap]; hg_pack2s = [codes][run table][bitmap, rebuilt by expand_runs_kernel]
        const size_t up = it.kind == KIND_PACK2 ? cbytes + mbytes : it.blob_bytes;
        const size_t blob = it.kind == KIND_PACK2 ? up : it.blob_bytes + mbytes;
        if (c.pk_bytes + blob + 64 > c.pk_cap) {  // (+ the readable slack behind the last blob) the packed area grows between chunks' uses (nothing of this chunk is in flight
          // unless earlier genomes of it are: wait for their copies before the old block goes away)
          ST_HIP(s, hipStreamSynchronize(e.copy));
          ST_HIP(s, hipStreamSynchronize(e.copy2));
          uint8_t *nb = nullptr;
          const size_t want = std::max(c.pk_bytes + blob + blob / 8 + 64, (size_t)(CHUNK_BYTES * 3 / 2 + (1u << 20)));  // (a chunk of blobs uploads up to CHUNK_BYTES; sparse ones add their rebuilt bitmaps)
          hipError_t he = hipMalloc(reinterpret_cast<void **>(&nb), want);
          if (he != hipSuccess) {
            fail(s, HG_ERR_OOM, "hipMalloc(" + std::to_string(want) + "): " + hipGetErrorString(he));
            return false;
          }
          if (c.dpk && c.pk_bytes) ST_HIP(s, hipMemcpy(nb, c.dpk, c.pk_bytes, hipMemcpyDeviceToDevice));
          if (c.dpk) ST_HIP(s, hipFree(c.dpk));
          c.dpk = nb, c.pk_cap = want;
        }
        if (!c.h_jobs) {
          ST_HIP(s, hipHostMalloc(reinterpret_cast<void **>(&c.h_jobs), CHUNK_GENOMES * sizeof(UnpackJob), hipHostMallocDefault));
          ST_HIP(s, hipMalloc(reinterpret_cast<void **>(&c.d_jobs), CHUNK_GENOMES * sizeof(UnpackJob)));
          ST_HIP(s, hipHostMalloc(reinterpret_cast<void **>(&c.h_sjobs), CHUNK_GENOMES * sizeof(SparseJob), hipHostMallocDefault));
          ST_HIP(s, hipMalloc(reinterpret_cast<void **>(&c.d_sjobs), CHUNK_GENOMES * sizeof(SparseJob)));
        }
        ST_HIP(s, hipMemcpyAsync(c.dpk + c.pk_bytes, it.seq, up, hipMemcpyHostToDevice, (e.turn++ & 1u) ? e.copy2 : e.copy));
        const uint64_t mask_off = c.pk_bytes + (blob - mbytes);
        UnpackJob &jb = c.h_jobs[c.n_jobs++];
        jb.pk_off = c.pk_bytes, jb.out_off = c.bytes, jb.n_bps = it.len, jb.mask_off = mask_off, jb.first_block = c.n_blocks, jb.pad = 0;
        c.n_blocks += (uint32_t)(((it.len + 15) / 16 + UNPACK_GROUPS_PER_BLOCK - 1) / UNPACK_GROUPS_PER_BLOCK);
        if (it.kind == KIND_PACK2S) {
          SparseJob &sj = c.h_sjobs[c.n_sjobs++];
          sj.codes_off = c.pk_bytes, sj.mask_off = mask_off, sj.n_bps = it.len, sj.first_block = c.n_sblocks, sj.pad = 0;
          c.n_sblocks += (uint32_t)((mbytes / 4 + SLICE_WORDS - 1) / SLICE_WORDS);
        }
        c.pk_offs.push_back(c.pk_bytes), c.mask_offs.push_back(mask_off);
        c.pk_bytes += blob;
      } else if (it.len) {
        if (it.len < SMALL_BYTES && c.bytes + padded <= CHUNK_BYTES) {
          if (!c.stage) ST_HIP(s, hipHostMalloc(reinterpret_cast<void **>(&c.stage), CHUNK_BYTES, hipHostMallocDefault));
          if (c.run_hi == c.run_lo) c.run_lo = c.run_hi = c.bytes;
          std::memcpy(c.stage + c.bytes, it.seq, it.len);
          if (padded > it.len) std::memset(c.stage + c.bytes + it.len, 0, padded - it.len);
          c.run_hi = c.bytes + padded;
        } else {
          if (!flush_run(s, e, c)) return false;
          ST_HIP(s, hipMemcpyAsync(c.d + c.bytes, it.seq, it.len, hipMemcpyHostToDevice, (e.turn++ & 1u) ? e.copy2 : e.copy));
        }
      }
      e.t_up_copy += now_s() - tc;
      c.offs.push_back(c.bytes), c.lens.push_back(it.len), c.tags.push_back(it.tag);
      c.bytes += padded, c.link_bytes += link;
      // hand the chunk on when it is full -- or when nothing else is waiting: the kernels start at once and the
      // next genome opens a new chunk.  (A/B: keeping the chunk open while the kernels are busy halves the number of
      // chunks and is 7-10 % slower end to end -- results come back later, the readers' buffers free up later.)
      if (idle_after || (c.has_ascii ? c.bytes : c.link_bytes) >= limit || c.tags.size() >= CHUNK_GENOMES) {
        if (!hand_over(s, e, cur)) return false;
        cur = -1, limit = std::min(CHUNK_BYTES, 2 * limit);
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

void computer(hg_sketch_stream *s, Engine *ep, int wi) {
  Engine &e = *ep;
  Engine::Worker &w = e.w[wi];
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
      ST_HIP(s, hipStreamWaitEvent(w.ctx->stream, c.uploaded, 0));
      // (empty genomes of an otherwise packed chunk have no blob: offset 0, length 0 -- nothing is read for them)
      bool all_packed = c.n_jobs > 0;
      if (all_packed && c.pk_offs.size() != m) {
        size_t nonempty = 0;
        for (size_t g = 0; g < m; ++g) nonempty += c.lens[g] != 0;
        all_packed = nonempty == c.pk_offs.size();
      }
      hg_status st;
      if (c.n_sjobs) {  // the bitmaps of the genomes that came as codes + run table
        hipLaunchKernelGGL(expand_runs_kernel, dim3(c.n_sblocks), dim3(256), 0, w.ctx->stream, c.dpk, c.d_sjobs, c.n_sjobs);
        ST_HIP(s, hipGetLastError());
      }
      if (all_packed) {
        std::vector<uint64_t> po(m, 0), mo(m, 0);
        for (size_t g = 0, k = 0; g < m; ++g)
          if (c.lens[g]) po[g] = c.pk_offs[k], mo[g] = c.mask_offs[k], ++k;
        st = hg_sketch_batch_dev_packed_masks(w.ctx, c.dpk, po.data(), mo.data(), c.lens.data(), m, &s->p, w.d_hv, w.d_n2, w.d_nh);
      } else {
        if (c.n_jobs) {
          hipLaunchKernelGGL(unpack2_kernel, dim3(c.n_blocks), dim3(256), 0, w.ctx->stream, c.dpk, c.d, c.d_jobs, c.n_jobs);
          ST_HIP(s, hipGetLastError());
        }
        st = hg_sketch_batch_dev(w.ctx, c.d, c.offs.data(), c.lens.data(), m, &s->p, w.d_hv, w.d_n2, w.d_nh);
      }
      if (st != HG_OK) {
        fail(s, st, std::string("device ") + std::to_string(e.device) + ": " + hg_last_error(w.ctx));
        return false;
      }
      const size_t hvb = m * D * sizeof(int16_t), hvb_al = (CHUNK_GENOMES * D * sizeof(int16_t) + 63) & ~(size_t)63;
      ST_HIP(s, hipMemcpyAsync(w.h_res, w.d_hv, hvb, hipMemcpyDeviceToHost, w.ctx->stream));
      ST_HIP(s, hipMemcpyAsync(w.h_res + hvb_al, w.d_n2, m * 4, hipMemcpyDeviceToHost, w.ctx->stream));
      ST_HIP(s, hipMemcpyAsync(w.h_res + hvb_al + CHUNK_GENOMES * 4, w.d_nh, m * 4, hipMemcpyDeviceToHost, w.ctx->stream));
      ST_HIP(s, hipStreamSynchronize(w.ctx->stream));
      {
        // the step's check word: a chunk with a genome that outgrew its hit region is sketched again (synchronous path),
        // and the rows copied above are fetched once more
        bool redone = false;
        st = hg_sketch_resolve(w.ctx, &redone);
        if (st != HG_OK) {
          fail(s, st, std::string("device ") + std::to_string(e.device) + ": " + hg_last_error(w.ctx));
          return false;
        }
        if (redone) {
          ST_HIP(s, hipMemcpyAsync(w.h_res, w.d_hv, hvb, hipMemcpyDeviceToHost, w.ctx->stream));
          ST_HIP(s, hipMemcpyAsync(w.h_res + hvb_al, w.d_n2, m * 4, hipMemcpyDeviceToHost, w.ctx->stream));
          ST_HIP(s, hipMemcpyAsync(w.h_res + hvb_al + CHUNK_GENOMES * 4, w.d_nh, m * 4, hipMemcpyDeviceToHost, w.ctx->stream));
          ST_HIP(s, hipStreamSynchronize(w.ctx->stream));
        }
      }
      Done d;
      d.tags = c.tags;
      d.hv.assign(reinterpret_cast<int16_t *>(w.h_res), reinterpret_cast<int16_t *>(w.h_res) + m * D);
      d.n2.assign(reinterpret_cast<int32_t *>(w.h_res + hvb_al), reinterpret_cast<int32_t *>(w.h_res + hvb_al) + m);
      d.nh.assign(reinterpret_cast<uint32_t *>(w.h_res + hvb_al + CHUNK_GENOMES * 4),
                  reinterpret_cast<uint32_t *>(w.h_res + hvb_al + CHUNK_GENOMES * 4) + m);
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
    for (auto &w : e->w)
      if (w.th.joinable()) w.th.join();
    if (!e->w[0].ctx) {  // never opened (bad device id): nothing to release, and no HIP call that would leave an error behind
      delete e;
      continue;
    }
    (void)hipSetDevice(e->device);
    if (e->copy) (void)hipStreamSynchronize(e->copy);
    if (e->copy2) (void)hipStreamSynchronize(e->copy2);
    for (Chunk &c : e->chunk) {
      if (c.d) (void)hipFree(c.d);
      if (c.stage) (void)hipHostFree(c.stage);
      if (c.dpk) (void)hipFree(c.dpk);
      if (c.h_jobs) (void)hipHostFree(c.h_jobs);
      if (c.d_jobs) (void)hipFree(c.d_jobs);
      if (c.h_sjobs) (void)hipHostFree(c.h_sjobs);
      if (c.d_sjobs) (void)hipFree(c.d_sjobs);
      if (c.uploaded) (void)hipEventDestroy(c.uploaded);
    }
    for (auto &w : e->w) {
      if (w.d_hv) (void)hipFree(w.d_hv);
      if (w.h_res) (void)hipHostFree(w.h_res);
    }
    if (e->copy) (void)hipStreamDestroy(e->copy);
    if (e->copy2) (void)hipStreamDestroy(e->copy2);
    if (e->copy2_done) (void)hipEventDestroy(e->copy2_done);
    for (auto &w : e->w)
      if (w.ctx) hg_ctx_destroy(w.ctx);
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
    hg_status st = hg_ctx_create(device_ids[i], &e->w[0].ctx);
    hipError_t he = hipSuccess;
    if (st == HG_OK) {
      if ((he = hipSetDevice(e->device)) == hipSuccess)
        he = hipStreamCreateWithFlags(&e->copy, hipStreamNonBlocking);
      if (he == hipSuccess) he = hipStreamCreateWithFlags(&e->copy2, hipStreamNonBlocking);
      if (he == hipSuccess) he = hipEventCreateWithFlags(&e->copy2_done, hipEventDisableTiming);
      for (int k = 0; k < N_CHUNKS && he == hipSuccess; ++k) {
        if ((he = hipMalloc(reinterpret_cast<void **>(&e->chunk[k].d), CHUNK_BYTES + 64)) == hipSuccess)
          e->chunk[k].cap = CHUNK_BYTES + 64, he = hipEventCreateWithFlags(&e->chunk[k].uploaded, hipEventDisableTiming);
        e->free_chunks.push_back(k);
      }
      for (int wi = 0; wi < N_WORKERS && he == hipSuccess && st == HG_OK; ++wi) {
        Engine::Worker &w = e->w[wi];
        if (wi && (st = hg_ctx_create(device_ids[i], &w.ctx)) != HG_OK) break;
        void *dres = nullptr;
        if ((he = hipMalloc(&dres, hvb_al + CHUNK_GENOMES * 8)) != hipSuccess) break;
        w.d_hv = static_cast<int16_t *>(dres);
        w.d_n2 = reinterpret_cast<int32_t *>(static_cast<uint8_t *>(dres) + hvb_al);
        w.d_nh = reinterpret_cast<uint32_t *>(w.d_n2 + CHUNK_GENOMES);
        he = hipHostMalloc(reinterpret_cast<void **>(&w.h_res), hvb_al + CHUNK_GENOMES * 8, hipHostMallocDefault);
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
    for (int wi = 0; wi < N_WORKERS; ++wi) e->w[wi].th = std::thread(computer, s, e, wi);
  }
  *out = s;
  return HG_OK;
}

static hg_status push_item(hg_sketch_stream *s, const uint8_t *seq, size_t len, uint64_t tag, int kind, bool wait, size_t given_bytes = ~(size_t)0);

extern "C" hg_status hg_sketch_stream_push(hg_sketch_stream *s, const uint8_t *seq, size_t len, uint64_t tag) {
  return push_item(s, seq, len, tag, KIND_ASCII, true);
}

extern "C" hg_status hg_sketch_stream_push_packed(hg_sketch_stream *s, const uint8_t *blob, size_t n_bps, uint64_t tag) {
  return push_item(s, blob, n_bps, tag, KIND_PACK2, true);
}

extern "C" hg_status hg_sketch_stream_push_packed_sparse(hg_sketch_stream *s, const uint8_t *blob, size_t blob_bytes, size_t n_bps, uint64_t tag) {
  return push_item(s, blob, n_bps, tag, KIND_PACK2S, true, blob_bytes);
}

extern "C" hg_status hg_sketch_stream_try_push(hg_sketch_stream *s, const uint8_t *data, size_t n_bps, uint64_t tag, int packed) {
  if (packed < 0 || packed > 2) return HG_ERR_INVALID;
  return push_item(s, data, n_bps, tag, packed, false);
}

extern "C" size_t hg_sketch_stream_max_pending(const hg_sketch_stream *s) { return s ? s->max_pending : 0; }

static hg_status push_item(hg_sketch_stream *s, const uint8_t *seq, size_t len, uint64_t tag, int kind, bool wait, size_t given_bytes) {
  if (!s || (len && !seq)) return HG_ERR_INVALID;
  size_t blob_bytes = 0;
  if (kind == KIND_PACK2S && len) {  // the blob says how long its run table is -- nothing of it is believed unchecked:
    // the count is read only if the caller's buffer reaches that far, the blob must fit the buffer, and the table has to be
    // what expand_runs_kernel's binary search assumes (ascending, disjoint, non-empty runs inside the sequence)
    if (len >= ((size_t)1 << 32)) return HG_ERR_UNSUPPORTED;
    const size_t tab_off = (((len + 3) / 4) + 15) & ~(size_t)15;
    if (given_bytes < tab_off + 8) return HG_ERR_INVALID;
    uint32_t n_runs;
    std::memcpy(&n_runs, seq + tab_off, 4);
    if ((size_t)n_runs > len) return HG_ERR_INVALID;
    blob_bytes = hg_pack2s_size(len, n_runs);
    if (blob_bytes > given_bytes) return HG_ERR_INVALID;
    uint64_t prev_end = 0;
    for (uint32_t r = 0; r < n_runs; ++r) {
      uint32_t st_len[2];
      std::memcpy(st_len, seq + tab_off + 8 + (size_t)8 * r, 8);
      if (st_len[1] == 0 || (r && st_len[0] < prev_end) || (uint64_t)st_len[0] + st_len[1] > len) return HG_ERR_INVALID;
      prev_end = (uint64_t)st_len[0] + st_len[1];
    }
  }
  std::unique_lock<std::mutex> lk(s->mu);
  if (s->finishing) return HG_ERR_INVALID;
  if (!wait && s->err == HG_OK && s->pushed - s->popped >= s->max_pending) return HG_ERR_CAPACITY;  // would block
  s->cv_room.wait(lk, [&] { return s->pushed - s->popped < s->max_pending || s->err != HG_OK; });
  if (s->err != HG_OK) return s->err;
  Engine *best = s->eng[0];
  for (Engine *e : s->eng)
    if (e->load < best->load) best = e;
  best->in.push_back(Item{seq, len, tag, kind, blob_bytes});
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

hipError_t hg_launch_pack2(hipStream_t st, const uint8_t *d_seq, const uint64_t *d_tab, uint32_t n, uint32_t blocks_max,
                           uint32_t u2t, uint8_t *d_blobs) {
  for (uint32_t g0 = 0; g0 < n; g0 += 65535) {
    const uint32_t m = std::min<uint32_t>(65535u, n - g0);
    hipLaunchKernelGGL(pack2_kernel, dim3(blocks_max, m), dim3(256), 0, st, d_seq, d_tab + 3 * (size_t)g0, u2t, d_blobs);
    const hipError_t e = hipGetLastError();
    if (e != hipSuccess) return e;
  }
  return hipSuccess;
}

extern "C" hg_status hg_unpack2_dev(hg_ctx *c, const uint8_t *d_blob, size_t n_bps, uint8_t *d_seq_out) {
  if (!c) return HG_ERR_INVALID;
  if (n_bps == 0) return HG_OK;
  if (!d_blob || !d_seq_out) return hg_fail(c, HG_ERR_INVALID, "NULL argument");
  if (((uintptr_t)d_blob | (uintptr_t)d_seq_out) & 15) return hg_fail(c, HG_ERR_INVALID, "hg_unpack2_dev: pointers must be 16-byte aligned");
  HG_ENTER(c);
  const uint64_t groups = (n_bps + 15) / 16;
  hipLaunchKernelGGL(unpack2_one_kernel, dim3((unsigned)((groups + UNPACK_GROUPS_PER_BLOCK - 1) / UNPACK_GROUPS_PER_BLOCK)),
                     dim3(256), 0, c->stream, d_blob, d_seq_out, (uint64_t)n_bps);
  HG_HIP(c, hipGetLastError());
  return HG_OK;
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
