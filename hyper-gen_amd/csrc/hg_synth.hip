// hg_synth.hip -- deterministic synthetic genomes generated directly in HBM (benchmark / test
// utility; the generator is the repo's own).  Counter based, so the CPU oracle's
// orc_synth_genome produces the same bytes independently.
#include "hg_internal.h"

namespace {
__device__ __forceinline__ uint64_t splitmix64(uint64_t x) {
  uint64_t z = x + 0x9e3779b97f4a7c15ull;
  z = (z ^ (z >> 30)) * 0xbf58476d1ce4e5b9ull;
  z = (z ^ (z >> 27)) * 0x94d049bb133111ebull;
  return z ^ (z >> 31);
}

// one thread = 32 consecutive bases (one root word)
__global__ __launch_bounds__(256) void synth_kernel(uint64_t first, uint64_t L, uint32_t cluster_size,
                                                    uint32_t ppm, uint64_t stride, uint8_t *__restrict__ out) {
  const uint64_t g = first + blockIdx.y;
  uint8_t *__restrict__ dst = out + (uint64_t)blockIdx.y * stride;
  const uint64_t wi = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;  // word index
  if (wi == 0) dst[0] = 'N';
  const uint64_t p0 = wi * 32;
  if (p0 >= L) return;
  const uint64_t c = g / cluster_size, m = g % cluster_size;
  const uint64_t key_root = splitmix64(0x48595045ull + c);
  const uint64_t key_mut = splitmix64(0x4d555441ull + g);
  const uint64_t thr = (m * (uint64_t)ppm * 4294967296ull) / 1000000ull;
  const uint64_t w = splitmix64(key_root + wi);
  const uint32_t lut = 0x54474341u;  // "ACGT"
  for (uint32_t b = 0; b < 32 && p0 + b < L; ++b) {
    uint32_t code = (uint32_t)(w >> (2 * b)) & 3u;
    if (thr) {
      const uint64_t u = splitmix64(key_mut + p0 + b);
      if ((u >> 32) < thr) code = (code + 1 + (uint32_t)((u & 0xffff) % 3)) & 3u;
    }
    dst[1 + p0 + b] = (uint8_t)(lut >> (8 * code));
  }
}
}  // namespace

extern "C" hg_status hg_synth_genomes_dev(hg_ctx *c, uint64_t first_genome, size_t n, uint64_t L,
                                          uint32_t cluster_size, uint32_t sub_ppm_per_member, uint64_t stride,
                                          uint8_t *d_out) {
  if (!c) return HG_ERR_INVALID;
  if (n == 0) return HG_OK;
  if (!d_out || cluster_size == 0 || stride < L + 1 || n > 65535) return hg_fail(c, HG_ERR_INVALID, "bad synth arguments");
  HG_ENTER(c);
  const uint64_t words = (L + 31) / 32;
  dim3 grid((unsigned)((words + 255) / 256 ? (words + 255) / 256 : 1), (unsigned)n);
  hipLaunchKernelGGL(synth_kernel, grid, dim3(256), 0, c->stream, first_genome, L, cluster_size,
                     sub_ppm_per_member, stride, d_out);
  HG_HIP(c, hipGetLastError());
  return HG_OK;
}
