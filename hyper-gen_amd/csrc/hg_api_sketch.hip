// hg_api_sketch.hip -- the C ABI of include/hypergen.h, part 2: the sketch path -- batch plans, hash + sample -> sort / unique ->
// encode over the genomes of a batch (device-resident, host-fed with uploads and host-side 2-bit packing under them, one
// genome per call), page-locked read buffers.  What src/sketch.rs:35-56 and src/sketch_cuda.rs:79-166 do per file.
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

#include "hg_sketch.h"

static hg_status sketch_batch_dev_impl(hg_ctx *c, const uint8_t *d_seq, const uint64_t *offsets, const uint64_t *lens, size_t n,
                                       const hg_sketch_params *p, int16_t *d_hv, int32_t *d_norm2, uint32_t *d_nhash, bool packed,
                                       const uint64_t *mask_offs = nullptr) {
  if (!c) return HG_ERR_INVALID;
  hg_status s = hg_check_sketch_params(c, p);
  if (s != HG_OK) return s;
  if (n == 0) return HG_OK;
  if (!d_seq || !offsets || !lens || !d_hv || !d_norm2 || !d_nhash) return hg_fail(c, HG_ERR_INVALID, "NULL argument");
  if (n > 0x7FFFFFFFull) return hg_fail(c, HG_ERR_UNSUPPORTED, "more than 2^31 genomes in one batch");
  HG_HIP(c, hipSetDevice(c->device));  // (no HG_ENTER: hg_sketch_step reads the previous step's check word behind its own launches)
  if (!packed && c->dbg_kmer_input == "packed") {
    // test hook: the batch arrived as ASCII -- pack it here and run the packed kernels on the blobs, so that every
    // ASCII entry point (and with it every parity test) can be driven through both input forms
    if ((s = hg_sketch_resolve(c)) != HG_OK) return s;
    std::vector<uint64_t> hook_offs(n);
    uint64_t total = 0;
    for (size_t g = 0; g < n; ++g) hook_offs[g] = total, total += hg_pack2_size(lens[g]);
    if ((s = hg_ensure(c, c->w_pk, total + 64)) != HG_OK) return s;
    if ((s = hg_pack_batch(c, d_seq, offsets, lens, n, p->norm_mode, static_cast<uint8_t *>(c->w_pk.p), hook_offs.data())) != HG_OK) return s;
    return hg_sketch_step(c, static_cast<const uint8_t *>(c->w_pk.p), hook_offs.data(), lens, n, p, d_hv, d_norm2, d_nhash, true, nullptr);
  }
  return hg_sketch_step(c, d_seq, offsets, lens, n, p, d_hv, d_norm2, d_nhash, packed, mask_offs);
}

hg_status hg_sketch_batch_dev_packed_masks(hg_ctx *c, const uint8_t *d_blobs, const uint64_t *code_offs, const uint64_t *mask_offs,
                                           const uint64_t *n_bps, size_t n, const hg_sketch_params *p, int16_t *d_hv,
                                           int32_t *d_norm2, uint32_t *d_nhash) {
  return sketch_batch_dev_impl(c, d_blobs, code_offs, n_bps, n, p, d_hv, d_norm2, d_nhash, true, mask_offs);
}

extern "C" hg_status hg_sketch_batch_dev(hg_ctx *c, const uint8_t *d_seq, const uint64_t *offsets,
                                         const uint64_t *lens, size_t n, const hg_sketch_params *p,
                                         int16_t *d_hv, int32_t *d_norm2, uint32_t *d_nhash) {
  return sketch_batch_dev_impl(c, d_seq, offsets, lens, n, p, d_hv, d_norm2, d_nhash, false);
}

extern "C" hg_status hg_sketch_batch_dev_packed(hg_ctx *c, const uint8_t *d_blobs, const uint64_t *offsets,
                                                const uint64_t *n_bps, size_t n, const hg_sketch_params *p,
                                                int16_t *d_hv, int32_t *d_norm2, uint32_t *d_nhash) {
  if (c && offsets)
    for (size_t g = 0; g < n; ++g)
      if (offsets[g] & 15) return hg_fail(c, HG_ERR_INVALID, "blob offsets must be multiples of 16");
  return sketch_batch_dev_impl(c, d_blobs, offsets, n_bps, n, p, d_hv, d_norm2, d_nhash, true);
}

extern "C" hg_status hg_pack2_batch_dev(hg_ctx *c, const uint8_t *d_seq, const uint64_t *offsets, const uint64_t *lens, size_t n,
                                        uint32_t norm_mode, uint8_t *d_blobs, const uint64_t *blob_offsets) {
  if (!c) return HG_ERR_INVALID;
  if (n == 0) return HG_OK;
  if (!d_seq || !offsets || !lens || !d_blobs || !blob_offsets || norm_mode > HG_NORM_U2T) return hg_fail(c, HG_ERR_INVALID, "bad argument");
  if (n > 0x7FFFFFFFull) return hg_fail(c, HG_ERR_UNSUPPORTED, "more than 2^31 genomes in one batch");
  HG_ENTER(c);
  return hg_pack_batch(c, d_seq, offsets, lens, n, norm_mode, d_blobs, blob_offsets);
}

extern "C" hg_status hg_pack2_dev(hg_ctx *c, const uint8_t *d_seq, size_t n_bps, uint32_t norm_mode, uint8_t *d_blob) {
  if (!c) return HG_ERR_INVALID;
  if (((uintptr_t)d_seq & 3) || ((uintptr_t)d_blob & 15)) return hg_fail(c, HG_ERR_INVALID, "hg_pack2_dev: d_seq must be 4-byte, d_blob 16-byte aligned");
  const uint64_t zero = 0, len = n_bps;
  return hg_pack2_batch_dev(c, d_seq, &zero, &len, 1, norm_mode, d_blob, &zero);
}

// NUMA node the device hangs off (sysfs of its PCI function), -1 when unknown.  Page-locked buffers filled by
// threads of that node are fetched ~25 % faster than buffers on the other socket (2-socket EPYC host, measured).
extern "C" int hg_device_numa_node(int device_id) {
  char bus[64] = {0};
  if (hipDeviceGetPCIBusId(bus, (int)sizeof bus, device_id) != hipSuccess) return -1;
  for (char *q = bus; *q; ++q) *q = (char)std::tolower((unsigned char)*q);
  const std::string path = std::string("/sys/bus/pci/devices/") + bus + "/numa_node";
  FILE *f = std::fopen(path.c_str(), "r");
  if (!f) return -1;
  int node = -1;
  if (std::fscanf(f, "%d", &node) != 1) node = -1;
  std::fclose(f);
  return node;
}

// ---- page-locked read buffers -----------------------------------------------------------------------------------
// A pageable hipMemcpyAsync goes through the runtime's bounce buffer and blocks its caller; sequence read straight
// into page-locked memory is DMA'd by hg_sketch_batch at the link rate instead.
namespace {
bool grow_pinned(uint8_t *&buf, size_t &cap, size_t need, size_t keep, void *) {
  if (need <= cap) return true;
  // recycled slots see files of similar but not equal sizes: round up so that they rarely move
  const size_t want = ((need + need / 8) + ((size_t)1 << 20) - 1) & ~(((size_t)1 << 20) - 1);
  void *nb = nullptr;
  if (hipHostMalloc(&nb, want, hipHostMallocPortable) != hipSuccess || !nb) return false;
  if (buf && keep) std::memcpy(nb, buf, std::min(keep, cap));
  if (buf) (void)hipHostFree(buf);
  buf = static_cast<uint8_t *>(nb), cap = want;
  return true;
}
}  // namespace

extern "C" hg_status hg_read_fastx_pinned(const char *path, uint32_t mode, uint8_t **buf, size_t *cap, size_t *n_bps) {
  return hg_read_fastx_impl(path, mode, buf, cap, n_bps, grow_pinned, nullptr);
}

extern "C" void hg_pinned_free(void *p) {
  if (p) (void)hipHostFree(p);
}

// Host-fed batch.  The batch is cut into sub-batches of about HG_STAGE_BYTES; a helper thread queues their
// uploads on the context's copy stream (one event per sub-batch) while this thread runs hash/sort/encode
// of the sub-batches already on the device, so PCIe transfer and kernels overlap for pinned and for
// pageable caller memory alike (a pageable hipMemcpyAsync blocks the thread that issues it).
constexpr uint64_t HG_STAGE_BYTES = 64ull << 20;
constexpr uint64_t HG_PACK_BYTES = HG_STAGE_BYTES + (2ull << 20);  // a sub-batch of genomes < 1 MiB each fits

// Host threads the library may use for its own host-side work on a call (2-bit packing of a host-fed batch): the cores
// this process may run on, at most 16 -- the reference's default `-t` (src/utils.rs:54-56).
static unsigned host_threads() {
  unsigned n = std::thread::hardware_concurrency();
  cpu_set_t set;
  if (sched_getaffinity(0, sizeof set, &set) == 0) n = std::min<unsigned>(n ? n : 1u, (unsigned)CPU_COUNT(&set));
  return std::max(1u, std::min(n, 16u));
}

// Host-fed calls in flight in this process (hg_sketch_batch / hg_kmer_hash_sample, any context): the reference's pattern is
// one call per genome from a pool of host threads (src/sketch_cuda.rs:79-96), and then the calls share ONE link.
static std::atomic<int> g_hostfed_calls{0};
namespace {
struct HostfedCall {
  int others;
  HostfedCall() : others(g_hostfed_calls.fetch_add(1)) {}
  ~HostfedCall() { g_hostfed_calls.fetch_sub(1); }
};
}  // namespace
static bool host_pinned(const void *p) {
  hipPointerAttribute_t a;
  if (hipPointerGetAttributes(&a, p) != hipSuccess) {
    (void)hipGetLastError();  // (an ordinary malloc'ed pointer is "invalid value" to the runtime)
    return false;
  }
  return a.type == hipMemoryTypeHost;
}
// Whether ONE genome handed over by a host-fed call goes over the link 2-bit packed (by the calling thread into the
// context's page-locked staging buffer) instead of as ASCII.
//  * pageable source (>= 256 KB): packed -- the runtime would stage it through its own pinned buffers anyway;
//  * page-locked source, fewer than 3 other host-fed calls in flight: ASCII (a lone 5 Mbp call takes 0.18 ms as ASCII,
//    0.33 ms packed);
//  * page-locked source, the link shared by K >= 4 calls (the reference's one-call-per-genome pattern from a thread
//    pool): packed as long as the host keeps up.  With K calls sharing a link of L bytes/s a call waits n K / L for its
//    ASCII, or n / r + 0.375 n K / L packed at r bytes/s: packing pays while r > L / (0.625 K).  r is what the calling
//    threads really achieve TOGETHER (16 of them are bound by host DRAM: 7 GB/s each on a quiet box of the pool -- 17 k
//    genomes/s against 10 k --, 4 GB/s on one whose memory was busy -- 8.6 k against 10 k), so it is measured in the
//    packed calls themselves (decayed mean, kept per range of K: r falls as K grows); when it falls short, the next calls
//    of that range go as ASCII -- 256 of them, doubling each time packing fails again, up to 16 384 -- and then packing is
//    tried afresh.
// Hook: "hostfed" = "ascii" never, "packed" always.
namespace {
constexpr double HG_LINK_BYTES_PER_S = 50e9;  // what ASCII uploads from page-locked memory reach on Gen5 x16 (bench.py host_fed.ascii_link)
struct PackState {
  std::atomic<uint64_t> rate{0};       // decayed mean of the bytes/s one calling thread packed at, contended packed calls
  std::atomic<uint32_t> samples{0};    // ... and how many calls it has seen since packing was (re)started
  std::atomic<int32_t> ascii_left{0};  // > 0: contended calls still to go as ASCII before packing is tried again
  std::atomic<uint32_t> backoff{256};
};
PackState g_pack[6];  // by calls in flight: 4-5, 6-7, 8-11, 12-15, 16-23, 24 and more
inline PackState &pack_state(int sharing) {
  return g_pack[sharing < 6 ? 0 : sharing < 8 ? 1 : sharing < 12 ? 2 : sharing < 16 ? 3 : sharing < 24 ? 4 : 5];
}
struct SingleChoice {
  bool packed = false, measured = false;
  int sharing = 1;  // calls in flight, this one included
};
// after a measured call packed its genome: n bytes in sec seconds
void pack_measured(const SingleChoice &ch, uint64_t n, double sec) {
  if (!ch.measured || sec <= 0) return;
  PackState &st = pack_state(ch.sharing);
  const uint64_t rate = (uint64_t)((double)n / sec), old = st.rate.load(std::memory_order_relaxed);
  const uint64_t now = old ? old - old / 8 + rate / 8 : rate;  // (racing updates lose a sample at worst)
  st.rate.store(now, std::memory_order_relaxed);
  const uint32_t seen = st.samples.fetch_add(1, std::memory_order_relaxed) + 1;
  if (seen >= 8 && (double)now * 0.625 * ch.sharing < 0.9 * HG_LINK_BYTES_PER_S) {
    const uint32_t b = st.backoff.load(std::memory_order_relaxed);
    st.ascii_left.store((int32_t)b, std::memory_order_relaxed);
    st.backoff.store(std::min<uint32_t>(2 * b, 16384u), std::memory_order_relaxed);
    st.samples.store(0, std::memory_order_relaxed), st.rate.store(0, std::memory_order_relaxed);
  } else if (seen == 1024) {
    st.backoff.store(256, std::memory_order_relaxed);  // (a long run of packing that paid)
  }
}
}  // namespace
static SingleChoice pack_single(const hg_ctx *c, const void *seq, uint64_t n_bps, int others) {
  SingleChoice r;
  r.sharing = others + 1;
  if (c->dbg_hostfed == "ascii" || hg_pack2_size(n_bps) > HG_PACK_BYTES) return r;
  r.packed = true;
  if (c->dbg_hostfed == "packed") return r;
  r.packed = false;
  if (n_bps < (256u << 10)) return r;
  if (!host_pinned(seq)) {
    r.packed = true;
    return r;
  }
  if (others < 3) return r;
  PackState &st = pack_state(r.sharing);
  if (st.ascii_left.load(std::memory_order_relaxed) > 0) {
    st.ascii_left.fetch_sub(1, std::memory_order_relaxed);
    return r;
  }
  r.packed = r.measured = true;
  return r;
}

// Page-locked staging buffer b of the context with room for `need` bytes, free to be rewritten (the upload that last read
// it has passed).  Sized by need -- locking pages costs ~0.25 ms per MB, and a pool of one-call-per-genome contexts
// should not pin 66 MB each for 2 MB blobs.
static hipError_t pack_buf_for(hg_ctx *c, int b, size_t need) {
  hipError_t e = hipSuccess;
  if (c->pack_used[b]) e = hipEventSynchronize(c->pack_ev[b]);
  if (e != hipSuccess) return e;
  if (!c->pack_ev[b] && (e = hipEventCreateWithFlags(&c->pack_ev[b], hipEventDisableTiming)) != hipSuccess) return e;
  if (c->pack_cap[b] >= need) return hipSuccess;
  if (c->pack_buf[b]) (void)hipHostFree(c->pack_buf[b]);
  c->pack_buf[b] = nullptr, c->pack_cap[b] = 0, c->pack_used[b] = false;
  const size_t want = (need + need / 4 + ((size_t)1 << 20) - 1) & ~(((size_t)1 << 20) - 1);
  if ((e = hipHostMalloc(&c->pack_buf[b], want, hipHostMallocDefault)) != hipSuccess) return e;
  c->pack_cap[b] = want;
  return hipSuccess;
}

extern "C" hg_status hg_sketch_batch(hg_ctx *c, const uint8_t *const *seqs, const size_t *lens, size_t n,
                                     const hg_sketch_params *p, int16_t *hv_out, int32_t *norm2_out,
                                     uint32_t *nhash_out) {
  if (!c) return HG_ERR_INVALID;
  hg_status s = hg_check_sketch_params(c, p);
  if (s != HG_OK) return s;
  if (n == 0) return HG_OK;
  if (!seqs || !lens || !hv_out || !norm2_out || !nhash_out) return hg_fail(c, HG_ERR_INVALID, "NULL argument");
  HG_ENTER(c);
  // The link is what limits this entry point (50 GB/s = 10 k genomes/s of 5 Mbp as ASCII): a batch that is worth it goes
  // over as 2-bit packed bases -- hg_pack2 blobs, 0.375 bytes per base, packed by a few host threads of this call into the
  // page-locked staging buffers while the previous sub-batch uploads -- and is sketched by the packed-input kernels
  // (bit-identical results).  Needs cores: with fewer than 4 usable ones the ASCII path stays.  Hook: "hostfed" = "ascii".
  // A call that hands over little (the n = 1 of the one-call-per-genome pattern) packs on its own thread, when
  // pack_single() says the link is contended.
  HostfedCall in_flight;
  unsigned P = host_threads();
  uint64_t all_bytes = 0;
  for (size_t g = 0; g < n; ++g) all_bytes += lens[g];
  // (batches of genomes below 1 kbp on average stay ASCII: hg_pack2 blobs carry 32 bytes of padding each)
  bool want_pack = (P >= 4 && all_bytes >= (32ull << 20) && all_bytes / n >= (1u << 10) && c->dbg_hostfed != "ascii") ||
                   (n > 1 && c->dbg_hostfed == "packed");
  if (want_pack) P = std::max(1u, P / (unsigned)(1 + in_flight.others));
  SingleChoice single;
  if (!want_pack && n == 1) {
    single = pack_single(c, seqs[0], lens[0], in_flight.others);
    if (single.packed) want_pack = true, P = 1;
  }
  const uint64_t stage_bytes = want_pack ? 2 * HG_STAGE_BYTES : HG_STAGE_BYTES;  // (packed: 48 MB per upload)
  // device layout: 16-byte aligned starts, 64 bytes of slack; sub-batch boundaries by bytes
  std::vector<uint64_t> offs(n), l64(n), boffs(n);
  std::vector<size_t> cut{0};
  uint64_t total = 0, in_chunk = 0;
  for (size_t g = 0; g < n; ++g) {
    if (lens[g] && !seqs[g]) return hg_fail(c, HG_ERR_INVALID, "NULL sequence");
    if (in_chunk >= stage_bytes) cut.push_back(g), in_chunk = 0;
    offs[g] = total, l64[g] = lens[g];
    const uint64_t padded = (lens[g] + 15) & ~(uint64_t)15;
    total += padded, in_chunk += padded;
  }
  cut.push_back(n);
  const size_t n_chunks = cut.size() - 1;
  // packed sub-batches: blob g of sub-batch k at the sub-batch's own start in the device buffer (its ASCII region is
  // larger than its blobs) + the sum of the blob sizes in front of it; a sub-batch whose blobs outgrow a staging buffer
  // (one huge genome) stays ASCII
  std::vector<uint8_t> sub_packed(n_chunks, 0);
  std::vector<uint64_t> sub_pk_bytes(n_chunks, 0);
  if (want_pack) {
    for (size_t k = 0; k < n_chunks; ++k) {
      uint64_t at = 0;
      for (size_t g = cut[k]; g < cut[k + 1]; ++g) boffs[g] = offs[cut[k]] + at, at += hg_pack2_size(lens[g]);
      sub_pk_bytes[k] = at;
      // ... and one whose blobs outgrow its own ASCII region (hg_pack2_size is 32 for 1..16 bases, their padded ASCII 16:
      // a sub-batch of very short sequences) stays ASCII too -- its blobs would run into the next sub-batch's region, or,
      // for the last one, past the end of the buffer
      const uint64_t span = offs[cut[k + 1] - 1] + ((lens[cut[k + 1] - 1] + 15) & ~(uint64_t)15) - offs[cut[k]];
      sub_packed[k] = at <= HG_PACK_BYTES && at <= span && at > 0;
    }
  }
  std::unique_ptr<CallPool> pool;
  // The pool's threads run on the NUMA node the sequences lie on (page-locked memory from the HIP runtime: the device's node,
  // like the staging buffers they write; packing from the other socket is ~1.5x slower), the uploader thread too.
  int pack_node = -1;
  if (want_pack && P > 1) {
    for (size_t g = 0; g < n && pack_node < 0; ++g)
      if (lens[g]) pack_node = hg_numa_node_of(seqs[g]);
    if (pack_node < 0) pack_node = hg_device_numa_node(c->device);
  }
  if (want_pack) pool.reset(new CallPool(P, pack_node));  // (takes the threads it can get)
  // packing has to outrun the link to be worth it from page-locked sources (ASCII goes at ~50 GB/s from those): the
  // uploader times its first packed sub-batch and leaves the rest as ASCII when the host is too slow for that
  const bool src_pinned = want_pack && n > 1 && host_pinned(seqs[0]);
  if ((s = hg_ensure(c, c->w_seq, total + 64)) != HG_OK) return s;
  const size_t hv_bytes = n * (size_t)p->hv_d * sizeof(int16_t);
  if ((s = hg_ensure(c, c->w_hv, hv_bytes + n * 8 + 64)) != HG_OK) return s;
  auto *d_seq = static_cast<uint8_t *>(c->w_seq.p);
  auto *d_hv = static_cast<int16_t *>(c->w_hv.p);
  auto *d_n2 = reinterpret_cast<int32_t *>(static_cast<uint8_t *>(c->w_hv.p) + ((hv_bytes + 15) & ~(size_t)15));
  auto *d_nh = reinterpret_cast<uint32_t *>(d_n2 + n);
  // one sub-batch (the n = 1 of a one-call-per-genome pool above all): its upload goes on the context's own stream, in
  // front of its kernels -- no second stream, no event to wait for
  const bool one_stream = n_chunks == 1;
  if (!one_stream && !c->copy_stream) HG_HIP(c, hipStreamCreateWithFlags(&c->copy_stream, hipStreamNonBlocking));
  const hipStream_t up_stream = one_stream ? c->stream : c->copy_stream;
  while (!one_stream && c->copy_events.size() < n_chunks) {
    hipEvent_t e;
    HG_HIP(c, hipEventCreateWithFlags(&e, hipEventDisableTiming));
    c->copy_events.push_back(e);
  }

  const uint64_t redone_before = c->n_redone_steps;  // (a sub-batch whose step is run again leaves stale rows in the copies queued behind it)
  std::mutex mu;
  std::condition_variable cv;
  size_t queued = 0;  // sub-batches whose uploads and event are queued
  hipError_t copy_err = hipSuccess;
  auto upload = [&](size_t first_chunk) {
    hipError_t e = hipSetDevice(c->device);
    if (n_chunks > 1) (void)hg_bind_thread_to_numa_node(pack_node, P);  // (the helper thread only, never the caller's)
    for (size_t k = first_chunk; k < n_chunks; ++k) {
      const size_t g0 = cut[k], g1 = cut[k + 1];
      const uint64_t span = offs[g1 - 1] + ((l64[g1 - 1] + 15) & ~(uint64_t)15) - offs[g0];
      if (sub_packed[k]) {
        // 2-bit pack the sub-batch into page-locked staging (all host threads of the call), then ONE upload
        const int b = (int)(k & 1);
        if (e == hipSuccess) e = pack_buf_for(c, b, n == 1 ? sub_pk_bytes[k] : HG_PACK_BYTES);
        if (e == hipSuccess) {
          auto *pin = static_cast<uint8_t *>(c->pack_buf[b]);
          // pieces of 1 Mbase, so that the threads finish together whatever the genome sizes
          constexpr uint64_t PIECE = 1ull << 20;
          std::vector<std::pair<size_t, uint64_t>> pieces;
          for (size_t g = g0; g < g1; ++g)
            for (uint64_t b = 0; b < lens[g]; b += PIECE) pieces.emplace_back(g, b);
          // ... handed out in runs of at least 256 kbase: a task per 5 kbp genome cost more in the pool's hand-overs than
          // in packing (100 000 x 5 kbp: 107 ms packed against 30 ms as ASCII through one staging copy)
          std::vector<size_t> task_first{0};
          {
            uint64_t in_task = 0;
            for (size_t i = 0; i < pieces.size(); ++i) {
              if (in_task >= (256u << 10)) task_first.push_back(i), in_task = 0;
              in_task += std::min<uint64_t>(lens[pieces[i].first] - pieces[i].second, PIECE);
            }
            task_first.push_back(pieces.size());
          }
          const auto t0 = std::chrono::steady_clock::now();
          pool->run(task_first.size() - 1, [&](size_t t) {
            for (size_t i = task_first[t]; i < task_first[t + 1]; ++i) {
              const size_t g = pieces[i].first;
              const uint64_t b = pieces[i].second;
              hg_pack2_piece(seqs[g], lens[g], p->norm_mode, pin + (boffs[g] - boffs[g0]), b, std::min<uint64_t>(lens[g], b + PIECE));
            }
          });
          const double sec = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
          // (page-locked sources go at ~55 GB/s as ASCII; pageable ones through the runtime's bounce buffer at ~12 GB/s: a
          // host whose cores are capped by a cgroup quota -- host_threads() cannot see one -- may pack slower than even that)
          if (n > 1 && k == first_chunk && (double)span < (src_pinned ? 55e9 : 12e9) * sec && c->dbg_hostfed != "packed")
            for (size_t j = k + 1; j < n_chunks; ++j) sub_packed[j] = 0;
          if (n == 1) pack_measured(single, lens[0], sec);
          if (e == hipSuccess) e = hipMemcpyAsync(d_seq + boffs[g0], pin, sub_pk_bytes[k], hipMemcpyHostToDevice, up_stream);
          if (e == hipSuccess) e = hipEventRecord(c->pack_ev[b], up_stream);
          c->pack_used[b] = true;
        }
      } else if (g1 - g0 >= 16 && span / (g1 - g0) < ((uint64_t)1 << 20) && span <= HG_PACK_BYTES) {
        // many small genomes: pack them into pinned memory (device layout) and upload once -- a
        // hipMemcpyAsync per 2 kbp genome costs more than the genome
        const int b = (int)(k & 1);
        if (e == hipSuccess) e = pack_buf_for(c, b, HG_PACK_BYTES);
        if (e == hipSuccess) {
          auto *pin = static_cast<uint8_t *>(c->pack_buf[b]);
          for (size_t g = g0; g < g1; ++g)
            if (lens[g]) std::memcpy(pin + (offs[g] - offs[g0]), seqs[g], lens[g]);
          e = hipMemcpyAsync(d_seq + offs[g0], pin, span, hipMemcpyHostToDevice, up_stream);
          if (e == hipSuccess) e = hipEventRecord(c->pack_ev[b], up_stream);
          c->pack_used[b] = true;
        }
      } else {
        for (size_t g = g0; g < g1 && e == hipSuccess; ++g)
          if (lens[g]) e = hipMemcpyAsync(d_seq + offs[g], seqs[g], lens[g], hipMemcpyHostToDevice, up_stream);
      }
      if (e == hipSuccess && !one_stream) e = hipEventRecord(c->copy_events[k], up_stream);
      std::lock_guard<std::mutex> lk(mu);
      if (e != hipSuccess) copy_err = e;
      queued = e == hipSuccess ? k + 1 : n_chunks;  // on error release the consumer, which then reports it
      cv.notify_all();
      if (e != hipSuccess) return;
    }
  };
  std::thread uploader;
  if (n_chunks > 1) uploader = std::thread(upload, (size_t)0);
  else upload(0);
  s = HG_OK;
  for (size_t k = 0; k < n_chunks && s == HG_OK; ++k) {
    {
      std::unique_lock<std::mutex> lk(mu);
      cv.wait(lk, [&] { return queued > k; });
      if (copy_err != hipSuccess) {
        s = hg_fail(c, HG_ERR_HIP, std::string("sequence upload: ") + hipGetErrorString(copy_err));
        break;
      }
    }
    const size_t g0 = cut[k], m = cut[k + 1] - g0;
    hipError_t e = one_stream ? hipSuccess : hipStreamWaitEvent(c->stream, c->copy_events[k], 0);
    if (e != hipSuccess) {
      s = hg_fail(c, HG_ERR_HIP, std::string("hipStreamWaitEvent: ") + hipGetErrorString(e));
      break;
    }
    if (sub_packed[k])
      s = hg_sketch_batch_dev_packed(c, d_seq, boffs.data() + g0, l64.data() + g0, m, p, d_hv + g0 * (size_t)p->hv_d, d_n2 + g0,
                                     d_nh + g0);
    else
      s = hg_sketch_batch_dev(c, d_seq, offs.data() + g0, l64.data() + g0, m, p, d_hv + g0 * (size_t)p->hv_d, d_n2 + g0,
                              d_nh + g0);
    if (s != HG_OK) break;
    e = hipMemcpyAsync(hv_out + g0 * (size_t)p->hv_d, d_hv + g0 * (size_t)p->hv_d, m * (size_t)p->hv_d * sizeof(int16_t),
                       hipMemcpyDeviceToHost, c->stream);
    if (e != hipSuccess) s = hg_fail(c, HG_ERR_HIP, std::string("hipMemcpyAsync: ") + hipGetErrorString(e));
  }
  if (uploader.joinable()) uploader.join();
  if (!one_stream) (void)hipStreamSynchronize(c->copy_stream);
  if (s != HG_OK) {
    (void)hipStreamSynchronize(c->stream);
    return s;
  }
  // the last sub-batch's check word (the earlier ones were read as their successors were queued)
  if ((s = hg_sketch_resolve(c)) != HG_OK) return s;
  if (c->n_redone_steps != redone_before)
    HG_HIP(c, hipMemcpyAsync(hv_out, d_hv, hv_bytes, hipMemcpyDeviceToHost, c->stream));
  HG_HIP(c, hipMemcpyAsync(norm2_out, d_n2, n * sizeof(int32_t), hipMemcpyDeviceToHost, c->stream));
  HG_HIP(c, hipMemcpyAsync(nhash_out, d_nh, n * sizeof(uint32_t), hipMemcpyDeviceToHost, c->stream));
  HG_HIP(c, hipStreamSynchronize(c->stream));
  return HG_OK;
}

extern "C" hg_status hg_kmer_hash_sample(hg_ctx *c, const uint8_t *seq, size_t n_bps, uint32_t ksize,
                                         uint64_t threshold, uint64_t seed, int canonical, uint32_t norm_mode,
                                         uint64_t *out_hashes, size_t cap, size_t *n_out) {
  if (!c) return HG_ERR_INVALID;
  if (!n_out) return hg_fail(c, HG_ERR_INVALID, "n_out == NULL");
  *n_out = 0;
  if (ksize < 1) return hg_fail(c, HG_ERR_INVALID, "ksize must be >= 1");
  if (ksize > 255) return hg_fail(c, HG_ERR_UNSUPPORTED, "ksize must be <= 255 (the reference's -k is u8)");
  if (norm_mode > HG_NORM_U2T) return hg_fail(c, HG_ERR_INVALID, "bad norm mode");
  if (n_bps && !seq) return hg_fail(c, HG_ERR_INVALID, "NULL sequence");
  if (n_bps < ksize) return HG_OK;
  HG_ENTER(c);
  const std::vector<uint64_t> offs{0}, l64{n_bps};
  hg_status s = hg_ensure(c, c->w_seq, n_bps + 64);
  if (s != HG_OK) return s;
  // over the link as ASCII, or 2-bit packed by this thread when the link is shared with other calls (pack_single())
  HostfedCall in_flight;
  const SingleChoice single = pack_single(c, seq, n_bps, in_flight.others);
  const bool packed = single.packed;
  if (packed) {
    HG_HIP(c, pack_buf_for(c, 0, hg_pack2_size(n_bps)));
    const auto tp0 = std::chrono::steady_clock::now();
    hg_pack2_piece(seq, n_bps, norm_mode, static_cast<uint8_t *>(c->pack_buf[0]), 0, n_bps);  // (the whole genome as one piece)
    pack_measured(single, n_bps, std::chrono::duration<double>(std::chrono::steady_clock::now() - tp0).count());
    HG_HIP(c, hipMemcpyAsync(c->w_seq.p, c->pack_buf[0], hg_pack2_size(n_bps), hipMemcpyHostToDevice, c->stream));
    HG_HIP(c, hipEventRecord(c->pack_ev[0], c->stream));
    c->pack_used[0] = true;
  } else {
    HG_HIP(c, hipMemcpyAsync(c->w_seq.p, seq, n_bps, hipMemcpyHostToDevice, c->stream));
  }
  // capacity heuristic wants "scaled"; derive it from the threshold (threshold = MAX / scaled)
  uint64_t scaled = threshold ? UINT64_MAX / threshold : UINT64_MAX;
  if (scaled < 1) scaled = 1;
  hg_batch_tables pl;
  uint32_t *d_nd = nullptr;
  hg_sample_fetch fetch;
  fetch.max_hashes = out_hashes ? cap : 0;
  s = hg_sample_batch_sync(c, static_cast<uint8_t *>(c->w_seq.p), offs.data(), l64.data(), 1, ksize, threshold, scaled,
                         seed, canonical != 0, norm_mode, pl, &d_nd, packed, nullptr, &fetch);
  if (s != HG_OK) return s;
  if (fetch.valid) {  // count and hashes came back with sample_batch's own synchronisation
    *n_out = fetch.nd;
    if (fetch.nd) std::memcpy(out_hashes, fetch.h_hashes, fetch.nd * sizeof(uint64_t));
    return HG_OK;
  }
  uint32_t nd = 0;
  HG_HIP(c, hipMemcpyAsync(&nd, d_nd, sizeof nd, hipMemcpyDeviceToHost, c->stream));
  HG_HIP(c, hipStreamSynchronize(c->stream));
  *n_out = nd;
  if (nd > cap) return hg_fail(c, HG_ERR_CAPACITY, "out_hashes too small");
  if (nd) {
    if (!out_hashes) return hg_fail(c, HG_ERR_INVALID, "out_hashes == NULL");
    HG_HIP(c, hipMemcpyAsync(out_hashes, static_cast<uint64_t *>(c->w_hits.p) + pl.meta[0].hit_off,
                             nd * sizeof(uint64_t), hipMemcpyDeviceToHost, c->stream));
    HG_HIP(c, hipStreamSynchronize(c->stream));
  }
  return HG_OK;
}

extern "C" hg_status hg_hv_encode(hg_ctx *c, const uint64_t *hashes, size_t n, uint32_t hv_d, uint32_t hv_layout,
                                  int16_t *hv_out, int32_t *norm2_out) {
  if (!c) return HG_ERR_INVALID;
  if (hv_d == 0 || hv_d > 32768) return hg_fail(c, HG_ERR_UNSUPPORTED, "hv_d must be in 1..32768");
  if (hv_layout > HG_LAYOUT_AVX2) return hg_fail(c, HG_ERR_INVALID, "bad layout");
  if ((n && !hashes) || !hv_out || !norm2_out) return hg_fail(c, HG_ERR_INVALID, "NULL argument");
  if (n > 0xFFFFFFF0ull) return hg_fail(c, HG_ERR_UNSUPPORTED, "too many hashes");
  HG_ENTER(c);
  hg_status s;
  if ((s = hg_ensure(c, c->w_gmeta, sizeof(hg_genome_meta))) != HG_OK) return s;
  if ((s = hg_ensure(c, c->w_hits, (n + 1) * sizeof(uint64_t))) != HG_OK) return s;
  if ((s = hg_ensure(c, c->w_cnt, 2 * sizeof(uint32_t))) != HG_OK) return s;
  if ((s = hg_ensure(c, c->w_hv, (size_t)hv_d * sizeof(int16_t) + 64)) != HG_OK) return s;
  c->plan.reset();  // w_gmeta is about to be overwritten
  hg_genome_meta m{};
  m.hit_off = 0, m.hit_cap = (uint32_t)n;
  const uint32_t nd = (uint32_t)n;
  auto *d_hv = static_cast<int16_t *>(c->w_hv.p);
  auto *d_n2 = reinterpret_cast<int32_t *>(static_cast<uint8_t *>(c->w_hv.p) + (((size_t)hv_d * 2 + 15) & ~(size_t)15));
  HG_HIP(c, hipMemcpyAsync(c->w_gmeta.p, &m, sizeof m, hipMemcpyHostToDevice, c->stream));
  HG_HIP(c, hipMemcpyAsync(c->w_cnt.p, &nd, sizeof nd, hipMemcpyHostToDevice, c->stream));
  if (n) HG_HIP(c, hipMemcpyAsync(c->w_hits.p, hashes, n * sizeof(uint64_t), hipMemcpyHostToDevice, c->stream));
  HG_HIP(c, hg_launch_encode(c->stream, static_cast<hg_genome_meta *>(c->w_gmeta.p), 1,
                             static_cast<uint64_t *>(c->w_hits.p), static_cast<uint32_t *>(c->w_cnt.p), hv_d,
                             hv_layout, d_hv, d_n2, nullptr, nd));
  HG_HIP(c, hipMemcpyAsync(hv_out, d_hv, (size_t)hv_d * sizeof(int16_t), hipMemcpyDeviceToHost, c->stream));
  HG_HIP(c, hipMemcpyAsync(norm2_out, d_n2, sizeof(int32_t), hipMemcpyDeviceToHost, c->stream));
  HG_HIP(c, hipStreamSynchronize(c->stream));
  return HG_OK;
}
