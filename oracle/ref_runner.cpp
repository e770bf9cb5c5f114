// ref_runner.cpp -- runs the REFERENCE's own kernel (src/cuda_kernel.cu, compiled in place by
// hipcc into oracle/_ref/ref_cuda_kernel.hsaco) on a sequence file and writes the resulting
// hash set.  TEST INFRASTRUCTURE ONLY: used to validate the CPU oracle and to produce golden
// vectors on the GPU box.  Host logic mirrors extract_kmer_t1ha2_cuda (src/sketch_cuda.rs:120-166):
// n_threads = ceil((n_bps-k+1)/512), block 1024 (cudarc LaunchConfig::for_num_elems), zeroed slot
// array, non-zero slots collected into a set.
//
//   ref_kmer_runner <hsaco> <seq.bin> <ksize> <scaled> <seed> <canonical 0|1> <slots_per_thread|0> <out.bin>
// slots_per_thread = 0 uses the reference's own value max(512/scaled*4, 8) (src/sketch_cuda.rs:136).
#include <hip/hip_runtime.h>

#include <algorithm>
#include <chrono>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <vector>

#define CK(x)                                                                      \
  do {                                                                             \
    hipError_t e = (x);                                                            \
    if (e != hipSuccess) {                                                         \
      std::fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e));                  \
      return 2;                                                                    \
    }                                                                              \
  } while (0)

int main(int argc, char **argv) {
  if (argc != 9) {
    std::fprintf(stderr, "usage: %s hsaco seq.bin ksize scaled seed canonical slots out.bin\n", argv[0]);
    return 1;
  }
  const char *hsaco = argv[1];
  FILE *f = std::fopen(argv[2], "rb");
  if (!f) return 1;
  std::fseek(f, 0, SEEK_END);
  const size_t n_bps = (size_t)std::ftell(f);
  std::fseek(f, 0, SEEK_SET);
  std::vector<uint8_t> seq(n_bps);
  if (n_bps && std::fread(seq.data(), 1, n_bps, f) != n_bps) return 1;
  std::fclose(f);
  size_t ksize = std::strtoull(argv[3], nullptr, 10);
  const uint64_t scaled = std::strtoull(argv[4], nullptr, 10);
  uint64_t seed = std::strtoull(argv[5], nullptr, 10);
  bool canonical = std::atoi(argv[6]) != 0;
  size_t slots = std::strtoull(argv[7], nullptr, 10);

  size_t kmer_per_thread = 512;
  if (n_bps < ksize) {
    FILE *o = std::fopen(argv[8], "wb");
    if (o) std::fclose(o);
    return 0;
  }
  const size_t n_kmers = n_bps - ksize + 1;
  const size_t n_threads = (n_kmers + kmer_per_thread - 1) / kmer_per_thread;
  if (slots == 0) slots = std::max<size_t>(kmer_per_thread / scaled * 4, 8);
  uint64_t threshold = UINT64_MAX / scaled;

  hipModule_t mod;
  hipFunction_t fn;
  CK(hipModuleLoad(&mod, hsaco));
  CK(hipModuleGetFunction(&fn, mod, "cuda_kmer_t1ha2"));
  uint8_t *d_seq;
  uint64_t *d_out;
  const size_t n_slots = slots * n_threads;
  CK(hipMalloc(&d_seq, n_bps + 64));
  CK(hipMalloc(&d_out, n_slots * sizeof(uint64_t)));
  CK(hipMemcpy(d_seq, seq.data(), n_bps, hipMemcpyHostToDevice));
  CK(hipMemset(d_out, 0, n_slots * sizeof(uint64_t)));
  void *args[] = {&d_seq, (void *)&n_bps, &kmer_per_thread, &slots, &ksize, &threshold, &seed, &canonical, &d_out};
  const unsigned block = 1024, grid = (unsigned)((n_threads + block - 1) / block);
  CK(hipModuleLaunchKernel(fn, grid, 1, 1, block, 1, 1, 0, nullptr, args, nullptr));  // warm-up + result
  CK(hipDeviceSynchronize());
  CK(hipMemset(d_out, 0, n_slots * sizeof(uint64_t)));
  auto t0 = std::chrono::steady_clock::now();
  CK(hipModuleLaunchKernel(fn, grid, 1, 1, block, 1, 1, 0, nullptr, args, nullptr));
  CK(hipDeviceSynchronize());
  const double ms = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count();
  std::vector<uint64_t> h(n_slots);
  CK(hipMemcpy(h.data(), d_out, n_slots * sizeof(uint64_t), hipMemcpyDeviceToHost));
  std::vector<uint64_t> set;
  for (uint64_t v : h)
    if (v != 0) set.push_back(v);  // src/sketch_cuda.rs:158-163
  std::sort(set.begin(), set.end());
  set.erase(std::unique(set.begin(), set.end()), set.end());
  FILE *o = std::fopen(argv[8], "wb");
  if (!o) return 1;
  if (!set.empty()) std::fwrite(set.data(), sizeof(uint64_t), set.size(), o);
  std::fclose(o);
  std::fprintf(stderr, "ref kernel: n_bps=%zu threads=%zu slots=%zu hashes=%zu kernel_ms=%.3f\n", n_bps, n_threads,
               slots, set.size(), ms);
  return 0;
}
