// Where do the ~80 ms between process start and "devices opened" go?  (development aid: hipcc --offload-arch=gfx950 -O2
// tools/hip_startup_probe.hip -o tools/_exp_startup -Lhyper-gen_amd -lhypergen_hip -Wl,-rpath,$PWD/hyper-gen_amd)
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>
#include <vector>
#include "../include/hypergen.h"
static double now() { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); }
__global__ void nop_kernel(int *p) { if (p) *p = 1; }
int main() {
  double t0 = now(), t;
#define STEP(what, expr) t = now(); expr; printf("%-52s %8.2f ms\n", what, (now() - t) * 1e3);
  STEP("hipInit(0)", (void)hipInit(0))
  int n = 0;
  STEP("hipGetDeviceCount", (void)hipGetDeviceCount(&n))
  STEP("hipSetDevice(0)", (void)hipSetDevice(0))
  STEP("hipFree(0) (context)", (void)hipFree(nullptr))
  hipStream_t s;
  STEP("hipStreamCreateWithFlags", (void)hipStreamCreateWithFlags(&s, hipStreamNonBlocking))
  void *d = nullptr, *h = nullptr;
  STEP("hipMalloc 1 MB (first)", (void)hipMalloc(&d, 1 << 20))
  STEP("hipHostMalloc 4 KB (first)", (void)hipHostMalloc(&h, 4096, hipHostMallocDefault))
  STEP("first kernel launch of this binary + sync", hipLaunchKernelGGL(nop_kernel, dim3(1), dim3(64), 0, s, (int *)d); (void)hipStreamSynchronize(s))
  STEP("second launch + sync", hipLaunchKernelGGL(nop_kernel, dim3(1), dim3(64), 0, s, (int *)d); (void)hipStreamSynchronize(s))
  hg_ctx *c = nullptr;
  STEP("hg_ctx_create(0) (after all of the above)", (void)hg_ctx_create(0, &c))
  void *d2 = nullptr;
  STEP("hg_dev_alloc 64 MB", (void)hg_dev_alloc(c, 64 << 20, &d2))
  std::vector<uint8_t> img(47 << 20, 1);
  STEP("hg_copy_h2d 47 MB from pageable memory (first)", (void)hg_copy_h2d(c, d2, img.data(), img.size()))
  STEP("hg_copy_h2d 47 MB from pageable memory (second)", (void)hg_copy_h2d(c, d2, img.data(), img.size()))
  // the library's code object: first launch of one of ITS kernels (hg_hv_unpack_batch_dev on one tiny payload)
  uint64_t off = 0; uint8_t q = 9, lay = 0;
  void *d3 = nullptr;
  (void)hg_dev_alloc(c, 1 << 20, &d3);
  STEP("first library kernel (hg_hv_unpack_batch_dev, 1 row)", (void)hg_hv_unpack_batch_dev(c, (const uint8_t *)d2, 47 << 20, &off, &q, &lay, 1, 4096, (int16_t *)d3))
  STEP("second library kernel", (void)hg_hv_unpack_batch_dev(c, (const uint8_t *)d2, 47 << 20, &off, &q, &lay, 1, 4096, (int16_t *)d3))
  std::vector<uint8_t> out(16 << 20);
  STEP("hg_copy_d2h 16 MB to fresh pageable memory (first)", (void)hg_copy_d2h(c, out.data(), d2, out.size()))
  STEP("hg_copy_d2h 16 MB to the same memory (second)", (void)hg_copy_d2h(c, out.data(), d2, out.size()))
  void *hp = nullptr;
  STEP("hipHostMalloc 16 MB", (void)hipHostMalloc(&hp, 16 << 20, hipHostMallocDefault))
  STEP("hg_copy_d2h 16 MB to pinned memory", (void)hg_copy_d2h(c, hp, d2, 16 << 20))
  printf("total %.1f ms\n", (now() - t0) * 1e3);
  return 0;
}
