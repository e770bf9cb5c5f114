// hg_cli.cpp -- `hyper-gen` command line on top of libhypergen_hip.so.
//
// Mirrors the reference's CLI surface (src/utils.rs:42-162, src/main.rs:11-24): subcommands
// sketch / dist / search with the same flags and defaults, the same .sketch container and the
// same ANI TSV (src/utils.rs:260-308).  All arithmetic runs on the MI355X through the C ABI;
// `-D cpu|gpu` only selects which of the reference's two base-normalisation behaviours is
// reproduced (cpu: needletail, u/U -> T; gpu: src/cuda_kernel.cu, ACGTacgt only).
#include <fcntl.h>
#include <glob.h>
#include <unistd.h>
#include <sched.h>
#include <unistd.h>
#include <sys/stat.h>

#include <algorithm>
#include <atomic>
#include <cerrno>
#include <chrono>
#include <condition_variable>
#include <cstdarg>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <ctime>
#include <deque>
#include <map>
#include <mutex>
#include <string>
#include <thread>
#include <vector>

#include "../../include/hypergen.h"

namespace {

void logline(const char *lvl, const std::string &msg) {
  char ts[32];
  std::time_t t = std::time(nullptr);
  std::strftime(ts, sizeof ts, "%Y-%m-%d-%H:%M:%S", std::localtime(&t));  // src/utils.rs:17-29
  std::printf("%s [%s] - %s\n", ts, lvl, msg.c_str());
  std::fflush(stdout);
}

// RUST_LOG=debug (the reference logs through env_logger) adds per-stage timings
bool debug_log() {
  static const bool on = [] {
    const char *e = std::getenv("RUST_LOG");
    return e && std::strstr(e, "debug") != nullptr;
  }();
  return on;
}
double now_s() { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); }
void debugf(const char *fmt, ...) __attribute__((format(printf, 1, 2)));
void debugf(const char *fmt, ...) {
  if (!debug_log()) return;
  char buf[512];
  va_list ap;
  va_start(ap, fmt);
  std::vsnprintf(buf, sizeof buf, fmt, ap);
  va_end(ap);
  logline("DEBUG", buf);
}

[[noreturn]] void die(const std::string &msg) {
  std::fprintf(stderr, "error: %s\n", msg.c_str());
  std::fflush(nullptr);
  _exit(2);  // not exit(): reader threads, or a thread that is bringing the HIP runtime up, may still be running
}

struct Cli {
  std::string mode, path = "1", path_r = "1", path_q = "1", out, method = "t1ha2", device = "cpu";
  bool pack_naive = false;  // --pack_layout naive: the payload layout of reference hosts without AVX2 (src/hd.rs:158-166)
  unsigned shards = 0;      // --shards N (dist / search; testing aid): N shards dealt round the visible devices instead of one each
  unsigned threads = 16, ksize = 21, top_n = 1;
  bool canonical = true;
  unsigned long long seed = 123, scaled = 1500, hv_d = 4096;
  float quant_scale = 1.0f, ani_th = 85.0f;
};

Cli parse(int argc, char **argv) {
  if (argc < 2) die("usage: hyper-gen <sketch|dist|search> [options]   (see --help)");
  Cli c;
  c.mode = argv[1];
  if (c.mode == "--version" || c.mode == "-V") {
    std::printf("hyper-gen 0.0.1 (%s)\n", hg_version());
    std::exit(0);
  }
  if (c.mode == "--help" || c.mode == "-h") {
    std::printf("HyperGen: Fast and memory-efficient genome sketching in hyperdimensional space (MI355X build)\n\n"
                "  hyper-gen sketch -p {fna_path} -o {output_sketch_file}\n"
                "  hyper-gen dist -r {ref_sketch} -q {query_sketch} -o {output_ANI_results}\n"
                "  hyper-gen search -r {ref_sketch} -q {query_sketch} -o {top_hits_per_query} [-n top_n]\n\n"
                "options: -p --path, -r --path_r, -q --path_q, -o --out, -t --thread [16], -m --sketch_method,\n"
                "         -C --canonical [true], -k --ksize [21], -S --seed [123], -s --scaled [1500], -d --hv_d [4096],\n"
                "         -Q --quant_scale [1.0], -a --ani_th [85.0], -D --device [cpu]\n"
                "extensions: -n --top_n [1] (search), --pack_layout avx2|naive [avx2] (sketch: the payload layout of\n"
                "         reference hosts with / without AVX2; dist and search read both), --shards N (dist / search: N\n"
                "         shards dealt round the visible GPUs; default one per GPU)\n");
    std::exit(0);
  }
  if (c.mode != "sketch" && c.mode != "dist" && c.mode != "search") die("unknown subcommand '" + c.mode + "'");
  if (c.mode != "sketch") c.method = "fracminhash";
  static const std::map<std::string, char> longs = {
      {"path", 'p'}, {"path_r", 'r'}, {"path_q", 'q'}, {"out", 'o'}, {"thread", 't'}, {"sketch_method", 'm'},
      {"canonical", 'C'}, {"ksize", 'k'}, {"seed", 'S'}, {"scaled", 's'}, {"hv_d", 'd'}, {"quant_scale", 'Q'},
      {"ani_th", 'a'}, {"device", 'D'}, {"top_n", 'n'}, {"pack_layout", 'L'}, {"shards", 'G'}};
  for (int i = 2; i < argc; ++i) {
    std::string a = argv[i], val;
    char key = 0;
    bool have_val = false;
    if (a.rfind("--", 0) == 0) {
      std::string name = a.substr(2);
      size_t eq = name.find('=');
      if (eq != std::string::npos) val = name.substr(eq + 1), name = name.substr(0, eq), have_val = true;
      auto it = longs.find(name);
      if (it == longs.end()) die("unexpected argument '" + a + "'");
      key = it->second;
    } else if (a.size() >= 2 && a[0] == '-') {
      key = a[1];
      if (a.size() > 2) val = a.substr(a[2] == '=' ? 3 : 2), have_val = true;
    } else {
      die("unexpected argument '" + a + "'");
    }
    if (!have_val) {
      if (i + 1 >= argc) die("a value is required for '" + a + "'");
      val = argv[++i];
    }
    auto u = [&](unsigned long long max) {
      char *e = nullptr;
      unsigned long long v = std::strtoull(val.c_str(), &e, 10);
      if (val.empty() || *e || v > max) die("invalid value '" + val + "' for '" + a + "'");
      return v;
    };
    switch (key) {
      case 'p': c.path = val; break;
      case 'r': c.path_r = val; break;
      case 'q': c.path_q = val; break;
      case 'o': c.out = val; break;
      case 't': c.threads = (unsigned)u(255); break;  // u8
      case 'm': c.method = val; break;
      case 'C':
        if (val == "true") c.canonical = true;
        else if (val == "false") c.canonical = false;
        else die("invalid value '" + val + "' for '--canonical'");
        break;
      case 'k': c.ksize = (unsigned)u(255); break;  // u8
      case 'S': c.seed = u(~0ull); break;
      case 's': c.scaled = u(~0ull); break;
      case 'd': c.hv_d = u(~0ull); break;
      case 'Q': c.quant_scale = std::strtof(val.c_str(), nullptr); break;
      case 'a': c.ani_th = std::strtof(val.c_str(), nullptr); break;
      case 'D': c.device = val; break;
      case 'n': c.top_n = (unsigned)u(1u << 20); break;  // search only (extension: the reference's search is a stub)
      case 'G': c.shards = (unsigned)u(64); break;  // dist / search only (testing aid: the several-GPU path on fewer GPUs)
      case 'L':  // sketch only (extension): which of the reference's two payload layouts to write
        if (val == "naive") c.pack_naive = true;
        else if (val == "avx2" || val == "bitpacker8x") c.pack_naive = false;
        else die("invalid value '" + val + "' for '--pack_layout' (avx2 | naive)");
        break;
      default: die("unexpected argument '" + a + "'");
    }
  }
  return c;
}

void ck(hg_ctx *ctx, hg_status s, const char *what) {
  if (s != HG_OK) die(std::string(what) + ": " + hg_status_str(s) + " (" + hg_last_error(ctx) + ")");
}
void ckm(hg_multi *m, hg_status s, const char *what) {
  if (s != HG_OK) die(std::string(what) + ": " + hg_status_str(s) + " (" + hg_multi_last_error(m) + ")");
}

// every GPU the process can see (HIP_VISIBLE_DEVICES narrows it); the reference opens device 0 only
// (src/sketch_cuda.rs:52)
hg_multi *open_all_devices(unsigned shards = 0) {
  const int n = hg_device_count();
  if (n <= 0) die(std::string("no MI355X device: ") + hg_last_error(nullptr));
  // (--shards N: N shards dealt round the devices -- repeated ids run several shards on one GPU, which is how the
  // several-GPU code path is exercised on a one-GPU box)
  std::vector<int> ids(shards ? shards : (unsigned)n);
  for (size_t i = 0; i < ids.size(); ++i) ids[i] = (int)(i % (size_t)n);
  hg_multi *m = nullptr;
  if (hg_multi_create(ids.data(), (int)ids.size(), &m) != HG_OK) die(std::string("no MI355X device: ") + hg_last_error(nullptr));
  return m;
}

// Reader thread -> the CPUs of the NUMA node its device hangs off (the thread's page-locked buffers lie there wherever
// the thread runs: filling and packing them from that socket is ~1.5x faster, and the DMA engine fetches them ~25 %
// faster than from the other one).  Best effort.
void bind_thread_to_node(int node, size_t threads_sharing) { (void)hg_bind_thread_to_numa_node(node, (unsigned)threads_sharing); }

// get_fasta_files (src/utils.rs:208-221): *.fna, *.fa, *.fasta, in that order
std::vector<std::string> fasta_files(const std::string &dir) {
  std::vector<std::string> out;
  for (const char *pat : {"*.fna", "*.fa", "*.fasta"}) {
    glob_t g;
    std::string p = dir + (dir.empty() || dir.back() == '/' ? "" : "/") + pat;
    if (glob(p.c_str(), 0, nullptr, &g) == 0)
      for (size_t i = 0; i < g.gl_pathc; ++i) out.push_back(g.gl_pathv[i]);
    globfree(&g);
  }
  return out;
}

int run_sketch(const Cli &c) {
  if (c.path == "1" && c.out.empty()) die("the following required arguments were not provided: --path --out");
  if (c.out.empty()) die("the following required arguments were not provided: --out");
  const auto files = fasta_files(c.path);
  const size_t n = files.size();
  logline("INFO", "Start sketching...");
  const auto t0 = std::chrono::steady_clock::now();
  if (c.scaled == 0) die("scaled must be >= 1");
  if (c.hv_d == 0 || c.hv_d > 32768) die("hv_d must be in 1..32768");
  if (c.hv_d % 256)  // the reference packs whole 256-blocks only (src/hd.rs:147) and says nothing; same bytes here
    logline("WARN", "hv_d is not a multiple of 256: the dimensions behind the last whole block are lost in the .sketch file (as in the reference)");
  hg_sketch_params p;
  hg_sketch_params_default(&p);
  const bool gpu_mode = c.device == "gpu";
  p.ksize = c.ksize, p.scaled = c.scaled, p.seed = c.seed;
  // -C is honoured by the reference's CUDA kernel only (src/cuda_kernel.cu:312-314); its CPU path always takes
  // needletail's canonical_kmers (src/sketch.rs:89) -- the flag still goes into the .sketch records as given
  p.canonical = gpu_mode ? (c.canonical ? 1u : 0u) : 1u;
  p.hv_d = (uint32_t)c.hv_d, p.hv_layout = HG_LAYOUT_AVX2;
  p.norm_mode = gpu_mode ? HG_NORM_ACGT : HG_NORM_U2T;
  const uint32_t read_mode = gpu_mode ? HG_READ_MERGE : HG_READ_NEEDLETAIL;

  std::vector<std::vector<int16_t>> payload(n);
  std::vector<hg_file_sketch> recs(n);

  // Reader side: -t persistent threads pull file indices from one counter; every thread owns S page-locked buffers
  // and fills them round robin (hg_read_fastx_pinned: the device then fetches the sequence by DMA at the PCIe rate
  // -- malloc'ed buffers went through the runtime's bounce buffer at a quarter of it), pushing each genome into the
  // sketch stream.  A thread waits when its S buffers are all in flight, so the page-locked footprint stays at T*S
  // genomes however many files there are (locking pages costs ~0.25 ms per MB) and after its first S files no
  // thread allocates.  The buffers are per thread on purpose: a ring shared by all readers made every file land in
  // memory last written on another core complex or socket -- 2x (16 threads) to 5x (32) slower reads on the
  // 2-socket host.
  // Device side: hg_sketch_stream (one uploader + one compute thread per visible GPU, genomes go to the least
  // loaded one: SURVEY.md 8e, no exchange); this thread collects the results, which arrive in completion order
  // and are stored by file index.
  const size_t T = std::max<size_t>(1, std::min<size_t>(c.threads, std::max<size_t>(n, 1)));
  const size_t S = std::max<size_t>(4, (64 + T - 1) / T);
  struct Slot {
    uint8_t *p = nullptr;
    size_t cap = 0, len = 0;
    bool busy = false;  // pushed, result not collected yet
  };
  std::vector<Slot> slots(T * S);
  std::vector<Slot *> slot_of(n, nullptr);
  std::mutex mu;
  std::condition_variable cv_stream;
  std::vector<std::condition_variable> cv_space(T);  // one per reader: a result wakes the reader that owns the buffer, not all sixteen
  hg_sketch_stream *stream = nullptr;
  std::atomic<size_t> next{0};
  std::atomic<uint64_t> read_ns{0};
  std::vector<int> dev_node;
  // 2-bit packing (hg_pack2: 3 bits per base over the link) costs a reader ~0.3 ms per 5 Mbp and only pays when
  // the link is what limits, so every reader decides per file: if at least two of its own buffers are still in
  // flight when it starts on a file, the device side lags behind the readers -> pack this one.
  std::atomic<size_t> n_packed{0};
  const uint32_t pack_flags = HG_READ_PACK2 | (p.norm_mode == HG_NORM_U2T ? HG_READ_PACK2_U2T : 0u);
  auto reader = [&](size_t tid) {
    bind_thread_to_node(dev_node[tid % dev_node.size()], T);
    size_t k = 0;
    for (size_t i; (i = next.fetch_add(1)) < n; k = (k + 1) % S) {
      Slot &sl = slots[tid * S + k];
      bool pack;
      {
        std::unique_lock<std::mutex> lk(mu);
        size_t in_flight = 0;
        for (size_t q = 0; q < S; ++q) in_flight += slots[tid * S + q].busy ? 1 : 0;
        pack = in_flight >= 2;
        cv_space[tid].wait(lk, [&] { return !sl.busy; });
      }
      const double tr0 = now_s();
      if (hg_read_fastx_pinned(files[i].c_str(), read_mode | (pack ? pack_flags : 0u), &sl.p, &sl.cap, &sl.len) != HG_OK)
        die("Opening .fna files failed: " + files[i]);
      read_ns.fetch_add((uint64_t)((now_s() - tr0) * 1e9));
      if (pack) n_packed.fetch_add(1);
      {
        std::unique_lock<std::mutex> lk(mu);
        sl.busy = true, slot_of[i] = &sl;
        cv_stream.wait(lk, [&] { return stream != nullptr; });  // the devices are opened while the first files are read
      }
      const hg_status ps = pack ? hg_sketch_stream_push_packed(stream, sl.p, sl.len, i) : hg_sketch_stream_push(stream, sl.p, sl.len, i);
      if (ps != HG_OK) die(std::string("sketch: ") + hg_sketch_stream_last_error(stream));
    }
  };
  const double td0 = now_s();
  const int nd = hg_device_count();  // (brings the HIP runtime up: the readers' page-locked buffers need it too)
  if (nd <= 0) die(std::string("no MI355X device: ") + hg_last_error(nullptr));
  for (int i = 0; i < nd; ++i) dev_node.push_back(hg_device_numa_node(i));
  std::vector<std::thread> readers;
  for (size_t t = 0; t < T && n; ++t) readers.emplace_back(reader, t);
  {
    std::vector<int> ids(nd);
    for (int i = 0; i < nd; ++i) ids[i] = i;
    hg_sketch_stream *st = nullptr;
    if (hg_sketch_stream_open(ids.data(), nd, &p, &st) != HG_OK) die(std::string("no MI355X device: ") + hg_last_error(nullptr));
    debugf("%d device(s) opened in %.1f ms", nd, (now_s() - td0) * 1e3);
    std::lock_guard<std::mutex> lk(mu);
    stream = st;
    cv_stream.notify_all();
  }
  std::vector<int16_t> hv(c.hv_d);
  double t_wait = 0, t_pack = 0;
  for (size_t done = 0; done < n; ++done) {
    uint64_t f = 0;
    int32_t n2 = 0;
    uint32_t nh = 0;
    int got = 0;
    const double tw0 = now_s();
    if (hg_sketch_stream_pop(stream, &f, hv.data(), &n2, &nh, &got) != HG_OK || !got)
      die(std::string("sketch: ") + hg_sketch_stream_last_error(stream));
    size_t owner;
    {
      std::lock_guard<std::mutex> lk(mu);
      slot_of[f]->busy = false;  // the sequence is not needed any more
      owner = (size_t)(slot_of[f] - slots.data()) / S;
    }
    cv_space[owner].notify_one();
    const double ts1 = now_s();
    const uint32_t q = hg_hv_quant_bits(hv.data(), (uint32_t)c.hv_d);  // if_compressed is hard-wired true (utils.rs:200)
    const size_t pk_bytes = c.pack_naive ? hg_hv_packed_bytes_naive((uint32_t)c.hv_d, q) : hg_hv_packed_bytes((uint32_t)c.hv_d, q);
    payload[f].resize((pk_bytes + 1) / 2);  // (the i16 view of the bytes, src/hd.rs:155-157)
    if ((c.pack_naive ? hg_hv_pack_naive : hg_hv_pack)(hv.data(), (uint32_t)c.hv_d, q, reinterpret_cast<uint8_t *>(payload[f].data())) != HG_OK) die("pack");
    hg_file_sketch &r = recs[f];
    std::memset(&r, 0, sizeof r);
    r.ksize = (uint8_t)c.ksize, r.canonical = c.canonical, r.hv_quant_bits = (uint8_t)q, r.hv_norm_2 = n2;
    r.scaled = c.scaled, r.seed = c.seed, r.hv_d = c.hv_d;
    r.file_str = files[f].c_str();
    r.hv = payload[f].data(), r.hv_len = pk_bytes / 2;  // align_to::<i16>().1: whole i16s
    t_wait += ts1 - tw0, t_pack += now_s() - ts1;
  }
  for (auto &t : readers) t.join();
  if (debug_log()) {
    double st[6];
    for (int e = 0; hg_sketch_stream_stats(stream, e, st) == HG_OK; ++e)
      debugf("device engine %d: uploader idle %.1f ms, waiting for a chunk %.1f ms, in copy calls %.1f ms; compute idle "
             "%.1f ms, busy %.1f ms; %.0f chunks", e, st[0] * 1e3, st[1] * 1e3, st[2] * 1e3, st[3] * 1e3, st[4] * 1e3, st[5]);
  }
  debugf("collector: waited %.1f ms for results, sketch compression %.1f ms; readers: %.2f ms per file and thread, "
         "%zu of %zu files sent 2-bit packed", t_wait * 1e3, t_pack * 1e3, n ? read_ns.load() / 1e6 / n : 0.0,
         n_packed.load(), n);
  const double secs = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
  char buf[256];
  std::snprintf(buf, sizeof buf, "Sketching %zu files took %.2fs - Speed: %.1f files/s", n, secs, n / std::max(secs, 1e-9));
  logline("INFO", buf);
  if (hg_sketch_file_write(c.out.c_str(), recs.data(), n) != HG_OK) die("Dump sketch file failed!");
  size_t total = 8;
  for (auto &r : recs) total += 47 + std::strlen(r.file_str) + r.hv_len * 2;
  std::snprintf(buf, sizeof buf, "Dump sketch file to %s with size %.2f MB", c.out.c_str(), total / 1024.0 / 1024.0);
  logline("INFO", buf);
  const double tf0 = now_s();
  hg_sketch_stream_close(stream);
  for (auto &sl : slots) hg_pinned_free(sl.p);
  debugf("released buffers and devices in %.1f ms", (now_s() - tf0) * 1e3);
  return 0;
}

// A loaded .sketch file: the records (names, norms, widths) on the host, the bit-packed payloads still inside the file
// image.  decompress_file_sketch (src/hd.rs:171-180) happens on the device: the image's payload bytes go over the link as
// they are (4.6 KB per sketch at 9 bits against 8 KB of int16) and hg_hv_unpack_batch_dev decodes them into the matrix
// the dist kernels read -- no host unpack threads, no second copy of the matrix in host memory.
struct Loaded {
  hg_sketch_file *f = nullptr;
  std::vector<int32_t> n2;
  std::vector<uint64_t> off;     // payload offsets in the image
  std::vector<uint8_t> q, lay;   // quantisation bits, payload layout (BitPacker8x / the non-AVX2 one) per record
  size_t n = 0;
  uint64_t hv_d = 0;
  uint8_t ksize = 0;
};

void load(const std::string &path, Loaded &L) {
  logline("INFO", "Loading sketch from " + path);
  if (hg_sketch_file_read_image(path.c_str(), &L.f) != HG_OK) die("Opening sketch file failed!");
  L.n = hg_sketch_file_count(L.f);
  if (L.n == 0) die("empty sketch file " + path);
  const hg_file_sketch *r0 = hg_sketch_file_get(L.f, 0);
  L.hv_d = r0->hv_d, L.ksize = r0->ksize;
  // validate before sizing anything from the file's own numbers
  if (L.hv_d == 0 || L.hv_d > 65536) die("unsupported HV dimension in " + path);
  L.n2.resize(L.n), L.off.resize(L.n), L.q.resize(L.n), L.lay.resize(L.n);
  for (size_t i = 0; i < L.n; ++i) {
    const hg_file_sketch *r = hg_sketch_file_get(L.f, i);
    if (r->hv_quant_bits < 1 || r->hv_quant_bits > 16) die("corrupt sketch record (quantisation bits) in " + path);
    if (r->hv_d != L.hv_d) die("sketches of one file use different HV dimensions");
    // which of the reference's two payload layouts this is follows from its length (they never coincide)
    const int lay = hg_hv_payload_layout((uint32_t)L.hv_d, r->hv_quant_bits, (size_t)r->hv_len * 2);
    if (lay < 0) die("corrupt sketch payload in " + path);
    L.n2[i] = r->hv_norm_2, L.off[i] = hg_sketch_file_payload_offset(L.f, i), L.q[i] = r->hv_quant_bits, L.lay[i] = (uint8_t)lay;
  }
}

// The sketches of a file on the devices: shard s (hg_shard_range) gets the slice of the file image that holds its
// records, decodes it there and keeps the int16 rows and the norms (what hg_dist_multi_dev takes).
struct DevSet {
  std::vector<const int16_t *> hv;
  std::vector<const int32_t *> n2;
  std::vector<size_t> rows;
};
void to_devices(hg_multi *m, const Loaded &L, DevSet &D) {
  char buf[96];
  std::snprintf(buf, sizeof buf, "Decompressing sketch with HV dim=%llu", (unsigned long long)L.hv_d);
  logline("INFO", buf);
  const int ns = hg_multi_size(m);
  D.hv.assign(ns, nullptr), D.n2.assign(ns, nullptr), D.rows.assign(ns, 0);
  size_t img_bytes = 0;
  const uint8_t *img = hg_sketch_file_image(L.f, &img_bytes);
  auto work = [&](int s) {
    size_t lo = 0, hi = 0;
    hg_shard_range(L.n, s, ns, &lo, &hi);
    if (hi == lo) return;
    hg_ctx *ctx = hg_multi_ctx(m, s);
    const uint64_t b0 = L.off[lo], b1 = L.off[hi - 1] + 2 * hg_sketch_file_get(L.f, hi - 1)->hv_len;
    if (b1 > img_bytes || b0 > b1) die("corrupt sketch payload");
    std::vector<uint64_t> rel(hi - lo);
    for (size_t i = lo; i < hi; ++i) rel[i - lo] = L.off[i] - b0;
    void *d_img = nullptr, *d_hv = nullptr, *d_n2 = nullptr;
    ck(ctx, hg_dev_alloc(ctx, b1 - b0, &d_img), "alloc");
    ck(ctx, hg_dev_alloc(ctx, (hi - lo) * L.hv_d * sizeof(int16_t), &d_hv), "alloc");
    ck(ctx, hg_dev_alloc(ctx, (hi - lo) * sizeof(int32_t), &d_n2), "alloc");
    ck(ctx, hg_copy_h2d(ctx, d_img, img + b0, b1 - b0), "upload");
    ck(ctx, hg_copy_h2d(ctx, d_n2, L.n2.data() + lo, (hi - lo) * sizeof(int32_t)), "upload");
    ck(ctx, hg_hv_unpack_batch_dev(ctx, static_cast<const uint8_t *>(d_img), b1 - b0, rel.data(), L.q.data() + lo, L.lay.data() + lo,
                                   hi - lo, (uint32_t)L.hv_d, static_cast<int16_t *>(d_hv)), "unpack");
    ck(ctx, hg_dev_free(ctx, d_img), "free");
    D.hv[s] = static_cast<const int16_t *>(d_hv), D.n2[s] = static_cast<const int32_t *>(d_n2), D.rows[s] = hi - lo;
  };
  std::vector<std::thread> th;
  for (int s = 1; s < ns; ++s) th.emplace_back(work, s);
  work(0);
  for (auto &t : th) t.join();
}
void release(hg_multi *m, DevSet &D) {
  for (size_t s = 0; s < D.hv.size(); ++s) {
    hg_ctx *ctx = hg_multi_ctx(m, (int)s);
    if (D.hv[s]) (void)hg_dev_free(ctx, const_cast<int16_t *>(D.hv[s]));
    if (D.n2[s]) (void)hg_dev_free(ctx, const_cast<int32_t *>(D.n2[s]));
  }
}

// "{:.3}" of an ANI (0 <= ani <= 100, src/utils.rs:277-282) without snprintf: ani * 1000 is exact in a double (24-bit
// significand times a 10-bit integer), so rounding that product to the nearest integer, ties to even (rint under the
// default rounding mode), is the correctly rounded decimal -- what Rust's exact-mode float formatting and glibc's %.3f
// both print.  Ties exist (ani = odd / 16: about one f32 in 16 000 near 96) and round-half-up would print 400 of the 1.1e9
// floats in [0, 100] differently: checked exhaustively against 128-bit integer arithmetic.  Writes "\t<int>.<3 digits>\n".
// (The library clamps ANI to [0, 100] -- src/dist.rs:156-159 -- so at most 3 + 1 + 3 digits follow the tab: the callers
// reserve 10 bytes per line for this.  The bound must not hang on another translation unit's arithmetic: anything that is
// not a number in [0, 100] -- a NaN, a negative value, a corrupted hit -- is brought into the range here.)
inline size_t put_ani(char *o, float ani) {
  if (!(ani >= 0.0f)) ani = 0.0f;  // NaN too
  if (ani > 100.0f) ani = 100.0f;
  const uint64_t v = (uint64_t)__builtin_rint((double)ani * 1000.0);
  uint64_t ip = v / 1000;
  const uint32_t fp = (uint32_t)(v % 1000);
  char tmp[24];
  size_t n = 0;
  do tmp[n++] = (char)('0' + ip % 10), ip /= 10;
  while (ip);
  size_t k = 0;
  o[k++] = '\t';
  while (n) o[k++] = tmp[--n];
  o[k++] = '.', o[k++] = (char)('0' + fp / 100), o[k++] = (char)('0' + fp / 10 % 10), o[k++] = (char)('0' + fp % 10), o[k++] = '\n';
  return k;
}

// An uninitialised array of hits (a std::vector would zero -- and so touch -- every page of a capacity-sized buffer of
// which a comparison fills one part in sixteen)
struct HitBuf {
  hg_ani_hit *p = nullptr;
  size_t n = 0;
  ~HitBuf() { std::free(p); }
  void resize(size_t m) {
    std::free(p);
    p = static_cast<hg_ani_hit *>(std::malloc(std::max<size_t>(m, 1) * sizeof(hg_ani_hit)));
    if (!p) die("out of memory");
    n = m;
  }
};

// All pairs with ANI >= ani_th of (R x Q) -- or of R against itself, i < j, when Q == nullptr --, on the host.  `order`:
// in dump_ani_file's order (src/utils.rs:262-269).  With one device the hits stay there until they are ordered (dist ->
// radix passes -> one download); with several, the shards' lists meet on the host and go through device 0 for the order.
// d_keep != nullptr: the list is wanted on device 0 (for hg_topk_per_query_dev), not on the host: *d_keep receives it
// (hg_dev_free it) and `hits` stays empty.
size_t all_hits(hg_multi *multi, const Loaded &R, const DevSet &dR, const Loaded *Q, const DevSet *dQ, float ani_th, bool order,
                HitBuf &hits, void **d_keep = nullptr) {
  const bool sym = Q == nullptr;
  const size_t qn = sym ? R.n : Q->n, total = sym ? R.n * (R.n - 1) / 2 : R.n * qn;
  size_t cap = std::max<size_t>(1024, total / 16), found = 0;
  const double t_in = now_s();
  if (hg_multi_size(multi) == 1) {
    hg_ctx *ctx = hg_multi_ctx(multi, 0);
    void *d_hits = nullptr;
    for (;;) {
      ck(ctx, hg_dev_alloc(ctx, cap * sizeof(hg_ani_hit), &d_hits), "alloc");
      const hg_status s = hg_dist_dev(ctx, dR.hv[0], dR.n2[0], R.n, sym ? dR.hv[0] : dQ->hv[0], sym ? dR.n2[0] : dQ->n2[0], qn,
                                      (uint32_t)R.hv_d, R.ksize, sym, ani_th, static_cast<hg_ani_hit *>(d_hits), cap, &found);
      if (s != HG_ERR_CAPACITY) {
        ck(ctx, s, "dist");
        break;
      }
      ck(ctx, hg_dev_free(ctx, d_hits), "free");
      cap = found;
    }
    const double t1 = now_s();
    if (order) ck(ctx, hg_sort_ani_hits_dev(ctx, static_cast<hg_ani_hit *>(d_hits), found, qn), "sort");
    if (d_keep) {
      *d_keep = d_hits;
      debugf("  dist on the device %.1f ms", (t1 - t_in) * 1e3);
      return found;
    }
    if (order) ck(ctx, hg_ctx_sync(ctx), "sort");
    const double t2 = now_s();
    hits.resize(found);
    if (found) ck(ctx, hg_copy_d2h(ctx, hits.p, d_hits, found * sizeof(hg_ani_hit)), "download");
    ck(ctx, hg_dev_free(ctx, d_hits), "free");
    debugf("  dist on the device %.1f ms, order %.1f ms, download %.1f ms", (t1 - t_in) * 1e3, (t2 - t1) * 1e3, (now_s() - t2) * 1e3);
    return found;
  }
  for (;;) {
    hits.resize(cap);
    // reference rows are all-gathered over xGMI, query rows stay on their shard's GPU
    const hg_status s = hg_dist_multi_dev(multi, dR.hv.data(), dR.n2.data(), dR.rows.data(), sym ? nullptr : dQ->hv.data(),
                                          sym ? nullptr : dQ->n2.data(), sym ? nullptr : dQ->rows.data(), (uint32_t)R.hv_d, R.ksize,
                                          sym, ani_th, hits.p, cap, &found);
    if (s != HG_ERR_CAPACITY) {
      ckm(multi, s, "dist");
      break;
    }
    cap = found;
  }
  hits.n = found;
  hg_ctx *ctx0 = hg_multi_ctx(multi, 0);
  if (order) ck(ctx0, hg_sort_ani_hits_staged(ctx0, hits.p, found, qn), "sort");
  if (d_keep) {
    ck(ctx0, hg_dev_alloc(ctx0, std::max<size_t>(found, 1) * sizeof(hg_ani_hit), d_keep), "alloc");
    if (found) ck(ctx0, hg_copy_h2d(ctx0, *d_keep, hits.p, found * sizeof(hg_ani_hit)), "upload");
  }
  return found;
}

int run_dist(const Cli &c) {
  if (c.path_r == "1" || c.path_q == "1" || c.out.empty())
    die("the following required arguments were not provided: --path_r --path_q --out");
  const auto t0 = std::chrono::steady_clock::now();
  const bool sym = c.path_r == c.path_q;  // src/dist.rs:13
  Loaded R, Qs;
  double tp = now_s();
  hg_multi *multi = nullptr;
  std::thread opener([&] {  // the HIP runtime comes up (~0.2 s) while the sketch files are read and decompressed
    const double td = now_s();
    multi = open_all_devices(c.shards);
    debugf("devices opened in %.1f ms", (now_s() - td) * 1e3);
  });
  {  // two files are read and parsed side by side
    std::thread second;
    if (!sym) second = std::thread([&] { load(c.path_q, Qs); });
    load(c.path_r, R);
    if (second.joinable()) second.join();
  }
  debugf("sketch files loaded in %.1f ms", (now_s() - tp) * 1e3);
  opener.join();
  const Loaded &Q = sym ? R : Qs;
  if (R.ksize != Q.ksize) die("Ref and query sketches use different kmer sizes!");
  if (R.hv_d != Q.hv_d) die("Ref and query sketches use different HV dimensions!");
  tp = now_s();
  DevSet dR, dQ;
  to_devices(multi, R, dR);
  if (!sym) to_devices(multi, Qs, dQ);
  debugf("payloads uploaded and decompressed on the device(s) in %.1f ms", (now_s() - tp) * 1e3);
  logline("INFO", "Computing ANI..");
  tp = now_s();
  const size_t total = sym ? R.n * (Q.n - 1) / 2 : R.n * Q.n;
  HitBuf hits;
  // (ordered on the device: dump_ani_file's order, src/utils.rs:262-269 -- two stable radix passes instead of a comparison
  // sort of up to 10^6..10^8 triples on one host core)
  const size_t found = all_hits(multi, R, dR, sym ? nullptr : &Qs, sym ? nullptr : &dQ, c.ani_th, true, hits);
  release(multi, dR), release(multi, dQ);
  debugf("ANI matrix (%zu hits), ordered, on the host in %.1f ms", found, (now_s() - tp) * 1e3);
  tp = now_s();
  // "{}\t{}\t{:.3}\n" (src/utils.rs:277-282), formatted by -t threads over contiguous ranges of the ordered hits: the
  // paths' lengths are looked up once per file, a line is two memcpy and put_ani
  const size_t FT = std::max<size_t>(1, std::min<size_t>(c.threads, found / 4096 + 1));
  std::vector<std::string> part(FT);
  {
    std::vector<uint32_t> len_r(R.n), len_q(Q.n);
    for (size_t i = 0; i < R.n; ++i) len_r[i] = (uint32_t)std::strlen(hg_sketch_file_get(R.f, i)->file_str);
    for (size_t i = 0; i < Q.n; ++i) len_q[i] = (uint32_t)std::strlen(hg_sketch_file_get(Q.f, i)->file_str);
    auto fmt = [&](size_t t) {
      const size_t lo = found * t / FT, hi = found * (t + 1) / FT;
      size_t need = 0;
      for (size_t i = lo; i < hi; ++i) need += (size_t)len_r[hits.p[i].ref_idx] + len_q[hits.p[i].qry_idx] + 10;
      std::string &o = part[t];
      o.resize(need);
      char *w = &o[0];
      for (size_t i = lo; i < hi; ++i) {
        const hg_ani_hit &h = hits.p[i];
        std::memcpy(w, hg_sketch_file_get(R.f, h.ref_idx)->file_str, len_r[h.ref_idx]);
        w += len_r[h.ref_idx];
        *w++ = '\t';
        std::memcpy(w, hg_sketch_file_get(Q.f, h.qry_idx)->file_str, len_q[h.qry_idx]);
        w += len_q[h.qry_idx];
        w += put_ani(w, h.ani);
      }
      o.resize((size_t)(w - &o[0]));
    };
    std::vector<std::thread> th;
    for (size_t t = 1; t < FT; ++t) th.emplace_back(fmt, t);
    fmt(0);
    for (auto &t : th) t.join();
  }
  size_t tsv_bytes = 0;
  for (const auto &o : part) tsv_bytes += o.size();
  debugf("TSV formatted (%.1f MB) in %.1f ms", tsv_bytes / 1e6, (now_s() - tp) * 1e3);
  tp = now_s();
  {  // every formatter's part goes to its own offset of the file: the copies into the page cache run side by side
    const int fd = ::open(c.out.c_str(), O_WRONLY | O_CREAT | O_TRUNC | O_CLOEXEC, 0666);
    if (fd < 0) die("Dump ANI file failed!");
    std::vector<size_t> at(FT + 1, 0);
    for (size_t t = 0; t < FT; ++t) at[t + 1] = at[t] + part[t].size();
    std::atomic<bool> bad{false};
    auto put = [&](size_t t) {
      const char *p = part[t].data();
      size_t left = part[t].size(), off = at[t];
      while (left) {
        const ssize_t w = ::pwrite(fd, p, left, (off_t)off);
        if (w < 0 && errno == EINTR) continue;
        if (w <= 0) {
          bad = true;
          return;
        }
        p += w, off += (size_t)w, left -= (size_t)w;
      }
    };
    std::vector<std::thread> th;
    for (size_t t = 1; t < FT; ++t) th.emplace_back(put, t);
    put(0);
    for (auto &t : th) t.join();
    if (::close(fd) != 0 || bad) die("Dump ANI file failed!");
  }
  debugf("TSV written in %.1f ms", (now_s() - tp) * 1e3);
  char buf[512];
  const double perc = total ? 100.0 * found / total : 0.0;
  if (perc < 5.0) {
    std::snprintf(buf, sizeof buf, "Output ANIs with threshold %.1f are too divergent: %zu of %zu (%.2f%%) ANIs are reported",
                  c.ani_th, found, total, perc);
    logline("WARN", buf);
  } else {
    std::snprintf(buf, sizeof buf, "Output %zu of %zu ANIs above threshold %.1f to file %s", found, total, c.ani_th, c.out.c_str());
    logline("INFO", buf);
  }
  std::snprintf(buf, sizeof buf, "Computed ANIs for %zu ref files and %zu query files took %.3fs", R.n, Q.n,
                std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count());
  logline("INFO", buf);
  hg_sketch_file_free(R.f);
  if (!sym) hg_sketch_file_free(Qs.f);
  hg_multi_destroy(multi);
  return 0;
}

// `search`: the reference parses the subcommand and does nothing (src/main.rs:22-24, "TODO: support for search").
// Here: every query sketch against the reference database, the top_n (-n, default 1) references per query with
// ANI >= ani_th, one line "query<TAB>reference<TAB>ani" per result, queries in file order, best first.
// Without -r / -q / -o it stays the reference's no-op.
int run_search(const Cli &c) {
  if (c.path_r == "1" || c.path_q == "1" || c.out.empty()) return 0;
  const auto t0 = std::chrono::steady_clock::now();
  Loaded R, Q;
  hg_multi *multi = nullptr;
  double tp = now_s();
  std::thread opener([&] {  // the HIP runtime comes up while the files are read
    const double td = now_s();
    multi = open_all_devices(c.shards);
    debugf("devices opened in %.1f ms", (now_s() - td) * 1e3);
  });
  {  // the two files are read and parsed side by side
    std::thread second([&] { load(c.path_q, Q); });
    load(c.path_r, R);
    second.join();
  }
  debugf("sketch files loaded in %.1f ms", (now_s() - tp) * 1e3);
  opener.join();
  if (R.ksize != Q.ksize) die("Ref and query sketches use different kmer sizes!");
  if (R.hv_d != Q.hv_d) die("Ref and query sketches use different HV dimensions!");
  tp = now_s();
  DevSet dR, dQ;
  to_devices(multi, R, dR);
  to_devices(multi, Q, dQ);
  debugf("payloads uploaded and decompressed on the device(s) in %.1f ms", (now_s() - tp) * 1e3);
  logline("INFO", "Searching..");
  tp = now_s();
  HitBuf hits;
  void *d_hits = nullptr, *d_out = nullptr, *d_cnt = nullptr;
  const size_t found = all_hits(multi, R, dR, &Q, &dQ, c.ani_th, false, hits, &d_hits);
  release(multi, dR), release(multi, dQ);
  debugf("ANI matrix (%zu hits) in %.1f ms", found, (now_s() - tp) * 1e3);
  tp = now_s();
  hg_ctx *ctx = hg_multi_ctx(multi, 0);
  const uint32_t k = std::max(1u, c.top_n);
  ck(ctx, hg_dev_alloc(ctx, Q.n * (size_t)k * sizeof(hg_ani_hit), &d_out), "alloc");
  ck(ctx, hg_dev_alloc(ctx, Q.n * sizeof(uint32_t), &d_cnt), "alloc");
  ck(ctx, hg_topk_per_query_dev(ctx, static_cast<hg_ani_hit *>(d_hits), found, Q.n, k, static_cast<hg_ani_hit *>(d_out),
                                static_cast<uint32_t *>(d_cnt)), "top-k");
  std::vector<hg_ani_hit> best(Q.n * (size_t)k);
  std::vector<uint32_t> cnt(Q.n);
  ck(ctx, hg_copy_d2h(ctx, best.data(), d_out, best.size() * sizeof(hg_ani_hit)), "download");
  ck(ctx, hg_copy_d2h(ctx, cnt.data(), d_cnt, cnt.size() * sizeof(uint32_t)), "download");
  hg_dev_free(ctx, d_hits), hg_dev_free(ctx, d_out), hg_dev_free(ctx, d_cnt);
  debugf("top-%u per query in %.1f ms", k, (now_s() - tp) * 1e3);
  tp = now_s();
  // "query<TAB>reference<TAB>ani" per result, queries in file order, best first; formatted by -t threads over contiguous
  // ranges of the queries (like dist's lines: two memcpy and put_ani per line)
  std::vector<size_t> first(Q.n + 1, 0);  // results in front of query q
  for (size_t q = 0; q < Q.n; ++q) first[q + 1] = first[q] + std::min<uint32_t>(cnt[q], k);
  const size_t reported = first[Q.n];
  const size_t FT = std::max<size_t>(1, std::min<size_t>(c.threads, reported / 4096 + 1));
  std::vector<std::string> part(FT);
  {
    std::vector<uint32_t> len_r(R.n), len_q(Q.n);
    for (size_t i = 0; i < R.n; ++i) len_r[i] = (uint32_t)std::strlen(hg_sketch_file_get(R.f, i)->file_str);
    for (size_t i = 0; i < Q.n; ++i) len_q[i] = (uint32_t)std::strlen(hg_sketch_file_get(Q.f, i)->file_str);
    auto fmt = [&](size_t t) {
      const size_t q_lo = Q.n * t / FT, q_hi = Q.n * (t + 1) / FT;
      size_t need = 0;
      for (size_t q = q_lo; q < q_hi; ++q)
        for (size_t r = 0; r < first[q + 1] - first[q]; ++r) need += (size_t)len_q[q] + len_r[best[q * k + r].ref_idx] + 10;
      std::string &o = part[t];
      o.resize(need);
      char *w = &o[0];
      for (size_t q = q_lo; q < q_hi; ++q) {
        const char *qs = hg_sketch_file_get(Q.f, q)->file_str;
        for (size_t r = 0; r < first[q + 1] - first[q]; ++r) {
          const hg_ani_hit &h = best[q * k + r];
          std::memcpy(w, qs, len_q[q]);
          w += len_q[q];
          *w++ = '\t';
          std::memcpy(w, hg_sketch_file_get(R.f, h.ref_idx)->file_str, len_r[h.ref_idx]);
          w += len_r[h.ref_idx];
          w += put_ani(w, h.ani);
        }
      }
      o.resize((size_t)(w - &o[0]));
    };
    std::vector<std::thread> th;
    for (size_t t = 1; t < FT; ++t) th.emplace_back(fmt, t);
    fmt(0);
    for (auto &t : th) t.join();
  }
  debugf("TSV formatted in %.1f ms", (now_s() - tp) * 1e3);
  size_t tsv_bytes = 0;
  FILE *f = std::fopen(c.out.c_str(), "wb");
  if (!f) die("Dump search file failed!");
  for (const auto &o : part) {
    if (o.size() && std::fwrite(o.data(), 1, o.size(), f) != o.size()) die("Dump search file failed!");
    tsv_bytes += o.size();
  }
  if (std::fclose(f) != 0) die("Dump search file failed!");
  debugf("TSV formatted and written (%.1f MB) in %.1f ms", tsv_bytes / 1e6, (now_s() - tp) * 1e3);
  char buf[256];
  std::snprintf(buf, sizeof buf, "Searched %zu queries against %zu references: %zu results (top %u, ANI >= %.1f) took %.3fs",
                Q.n, R.n, reported, k, c.ani_th, std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count());
  logline("INFO", buf);
  hg_sketch_file_free(R.f), hg_sketch_file_free(Q.f);
  hg_multi_destroy(multi);
  return 0;
}

}  // namespace

int main(int argc, char **argv) {
  const Cli c = parse(argc, argv);
  // (a normal return: leaving through _exit once the outputs are closed saves the runtime's exit handlers -- 20-40 ms of a
  // 10 000 x 10 000 dist -- but those handlers are also where rocprofv3 and other tools write what they collected)
  if (c.mode == "sketch") return run_sketch(c);
  if (c.mode == "dist") return run_dist(c);
  return run_search(c);
}
