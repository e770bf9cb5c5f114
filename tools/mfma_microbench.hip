// What the matrix pipe sustains in the shape of dist_mfma_kernel's main loop (256 x 320 tile: 8 waves of 80
// v_mfma_i32_16x16x64_i8 per K-step, one workgroup per CU), with the loop's other ingredients added one at a time:
//   mode 0  MFMAs on register operands only
//   mode 1  + the 19 ds_read_b128 fragment reads per K-step, one phase ahead, as in the kernel
//   mode 2  + one s_barrier per K-step
//   mode 3  + the LDS-DMA of a K-step (72 pieces of 1 KiB from four loader waves) out of a 2 MiB buffer (L2 resident)
//   mode 4  the same from a 2 GiB buffer (every workgroup streams its own rows: L2 misses)
// Every mode with constant operand bytes and with random ones: the matrix pipe's sustained rate depends on the data.
// Prints TFLOP/s (2 * 16 * 16 * 64 per instruction) and the fraction of the 5 PFLOP/s dense i8 peak.
//   hipcc --offload-arch=gfx950 -O3 tools/mfma_microbench.hip -o gpurun_out/mfma_microbench
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <cstdlib>
#include <type_traits>
#include <vector>

typedef int int4v __attribute__((ext_vector_type(4)));
typedef float float4v __attribute__((ext_vector_type(4)));
// DATA: 0 constant bytes, 1 random bytes (both v_mfma_i32_16x16x64_i8), 2 = e2m1 nibbles 0x2 / 0xA on v_mfma_scale_f32_16x16x128_f8f6f4
// (what the Hamming search feeds; operand bytes toggle in one bit per nibble only)
__host__ __device__ inline uint32_t nib(uint32_t x) { return 0x22222222u | ((x * 2654435761u ^ (x >> 7) * 40503u) & 0x88888888u); }
typedef __attribute__((address_space(3))) void *lds_ptr_t;
constexpr int STEPS = 512;              // K-steps per workgroup
constexpr int STAGE = (256 + 320) * 128;  // bytes of one operand stage

// MT = 16-row M tiles of this wave (8 in the kernel as shipped); ROW0 = its first row in the A stage
template <int MODE, int DATA, int MT, int NTN>
__device__ __forceinline__ void body(const uint8_t *__restrict__ src, size_t rows_stride, int *out, unsigned long long *clk, uint32_t arow0) {
  extern __shared__ __attribute__((aligned(16))) uint8_t lds[];
  const uint32_t tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  constexpr int MP = MT / 2, PH = 2 * MP;
  typedef typename std::conditional<DATA == 2, float4v, int4v>::type acc_t;
  acc_t acc[MT][NTN];
  const int fp4_scale = 0x7f7f7f7f;
#pragma unroll
  for (int m = 0; m < MT; ++m)
#pragma unroll
    for (int n = 0; n < NTN; ++n) acc[m][n] = acc_t{};
  int4v a[2][2], b[2][NTN];
#pragma unroll
  for (int i = 0; i < 2; ++i) {
    a[i][0] = a[i][1] = DATA == 2 ? int4v{(int)nib(tid), (int)nib(tid + 999), (int)nib(tid * 3), (int)nib(tid * 7 + 1)} : DATA ? int4v{(int)(tid * 2654435761u), (int)(tid * 40503u + 77), (int)(tid * 69069u), (int)(tid * 1664525u + 3)} : int4v{(int)tid, 1, 2, 3};
#pragma unroll
    for (int n = 0; n < NTN; ++n) b[i][n] = DATA == 2 ? int4v{(int)nib(lane + n), (int)nib(lane * 5 + n), (int)nib(tid + 31 * n), (int)nib(tid * 11 + n)} : DATA ? int4v{(int)(lane * 2246822519u + n), (int)(lane * 3266489917u), (int)(tid * 668265263u + n * 7), (int)(tid * 374761393u)} : int4v{(int)lane, n, 5, 7};
  }
  for (uint32_t i = tid; i < 2 * STAGE / 4; i += 512) reinterpret_cast<uint32_t *>(lds)[i] = DATA == 2 ? nib(i) : (DATA ? i * 2654435761u : 0x01010101u);
  __syncthreads();
  const uint32_t fr = lane & 15, fq = lane >> 4, wn = wave % 4;
  // the kernel's XOR swizzle: 16-byte chunk (kk * 4 + fq) ^ ((row >> 1) & 7) of a 128-byte row: conflict-free ds_read_b128
  const uint32_t swz = (fr >> 1) & 7;
  const uint32_t fa = (arow0 + fr) * 128 + (fq ^ swz) * 16, fb = 256 * 128 + (wn * (16 * NTN) + fr) * 128 + (fq ^ swz) * 16;
  const int32_t kk1_off = (int32_t)((((fq ^ swz) ^ 4) - (fq ^ swz)) * 16);
  __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(
      const_cast<uint8_t *>(src + (size_t)blockIdx.x * rows_stride), 0, 0x7fffffff, 0x00020000);
  const uint32_t voff = (tid & 255) * 16;
  const unsigned long long c0 = __builtin_amdgcn_s_memtime(), r0 = __builtin_amdgcn_s_memrealtime();
  for (int ks = 0; ks < STEPS; ++ks) {
    const uint8_t *st = lds + (ks & 1) * STAGE;
#pragma unroll
    for (int t = 0; t < PH; ++t) {  // PH phases of 10 MFMAs: (kk = t / MP, mp = t % MP)
      const int kk = t / MP, mp = t % MP;
      if (MODE >= 1 && t + 1 < PH) {
        const int kk1 = (t + 1) / MP, mp1 = (t + 1) % MP;
        if (mp1 == 0)
#pragma unroll
          for (int n = 0; n < NTN; ++n) b[kk1 & 1][n] = *reinterpret_cast<const int4v *>(st + fb + n * 16 * 128 + (kk1 ? kk1_off : 0));
#pragma unroll
        for (int i = 0; i < 2; ++i) a[(t + 1) & 1][i] = *reinterpret_cast<const int4v *>(st + fa + (2 * mp1 + i) * 16 * 128 + (kk1 ? kk1_off : 0));
      }
      if (t == PH - 1) {
        if (MODE >= 1) asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        if (MODE >= 2) __syncthreads();
        if (MODE >= 3 && wave < 4) {
#pragma unroll
          for (int i = 0; i < 18; ++i)
            __builtin_amdgcn_raw_ptr_buffer_load_lds(rs, (lds_ptr_t)(lds + (ks & 1) * STAGE + (wave * 18 + i) * 1024), 16,
                                                     voff + (uint32_t)i * 4096u, (MODE == 4 ? (ks & 15) * 131072 : 0), 0, 0);
        }
        if (MODE >= 1) {
          const uint8_t *nx = lds + ((ks + 1) & 1) * STAGE;
#pragma unroll
          for (int n = 0; n < NTN; ++n) b[0][n] = *reinterpret_cast<const int4v *>(nx + fb + n * 16 * 128);
#pragma unroll
          for (int i = 0; i < 2; ++i) a[0][i] = *reinterpret_cast<const int4v *>(nx + fa + i * 16 * 128);
        }
      }
      __builtin_amdgcn_sched_barrier(0);
#pragma unroll
      for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int n = 0; n < NTN; ++n)
          if constexpr (DATA == 2)
            asm volatile("v_mfma_scale_f32_16x16x128_f8f6f4 %0, %1, %2, %0, %3, %3 op_sel_hi:[0,0,0] cbsz:4 blgp:4"
                         : "+v"(acc[2 * mp + i][n]) : "v"(a[t & 1][i]), "v"(b[kk & 1][n]), "v"(fp4_scale));
          else acc[2 * mp + i][n] = __builtin_amdgcn_mfma_i32_16x16x64_i8(a[t & 1][i], b[kk & 1][n], acc[2 * mp + i][n], 0, 0, 0);
      __builtin_amdgcn_sched_barrier(0);
    }
  }
  if (tid == 0) clk[2 * blockIdx.x] = __builtin_amdgcn_s_memtime() - c0, clk[2 * blockIdx.x + 1] = __builtin_amdgcn_s_memrealtime() - r0;
  if constexpr (DATA == 2) asm volatile("s_nop 7\n\ts_nop 7\n\ts_nop 7" ::: "memory");
  int s = 0;
#pragma unroll
  for (int m = 0; m < MT; ++m)
#pragma unroll
    for (int n = 0; n < NTN; ++n) s += (int)acc[m][n][0] + (int)acc[m][n][1] + (int)acc[m][n][2] + (int)acc[m][n][3];
  out[blockIdx.x * 512 + tid] = s;
}
// SPLIT = M tiles of the four loader waves (waves 0..3, one per SIMD); their SIMD partners (waves 4..7) take 16 - SPLIT.
// 8 = the kernel as shipped.  A loader spends ~1 260 cycles of a K-step blocked in its 18 DMA issues and then still owes its
// own MFMAs: with fewer M tiles it catches up, its partner -- which only waits otherwise -- takes the difference.
template <int MODE, int DATA, int SPLIT = 8, int NTN = 5>
__global__ __launch_bounds__(512) void k(const uint8_t *__restrict__ src, size_t rows_stride, int *out, unsigned long long *clk) {
  if (SPLIT == 8 || threadIdx.x < 256) body<MODE, DATA, SPLIT, NTN>(src, rows_stride, out, clk, (threadIdx.x >> 8) * (SPLIT * 16));
  else body<MODE, DATA, 16 - SPLIT, NTN>(src, rows_stride, out, clk, SPLIT * 16);
}

template <int MODE, int DATA, int SPLIT = 8, int NTN = 5>
static void run(const uint8_t *src, size_t stride, int *out, int n_wg, const char *what) {
  static unsigned long long *clk = nullptr;
  if (!clk) hipMalloc(&clk, (size_t)n_wg * 16);
  const size_t lds = 2 * STAGE;
  hipFuncSetAttribute(reinterpret_cast<const void *>(&k<MODE, DATA, SPLIT, NTN>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
  hipEvent_t e0, e1;
  hipEventCreate(&e0), hipEventCreate(&e1);
  float best = 1e30f;
  for (int r = 0; r < 6; ++r) {
    hipEventRecord(e0);
    hipLaunchKernelGGL((k<MODE, DATA, SPLIT, NTN>), dim3(n_wg), dim3(512), lds, 0, src, stride, out, clk);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms;
    hipEventElapsedTime(&ms, e0, e1);
    if (r >= 2 && ms < best) best = ms;
  }
  std::vector<unsigned long long> h(2 * (size_t)n_wg);
  hipMemcpy(h.data(), clk, h.size() * 8, hipMemcpyDeviceToHost);
  double sc = 0, rt = 0;
  for (int i = 0; i < n_wg; ++i) sc += (double)h[2 * i], rt += (double)h[2 * i + 1];
  const double mhz = sc / rt * 100.0;  // s_memrealtime ticks at 100 MHz
  const double flops = 2.0 * 16 * 16 * (DATA == 2 ? 128 : 64) * (16.0 * NTN) * 8 * STEPS * n_wg;
  printf("mode %d data %s  %-62s %8.3f ms  %7.1f TFLOP/s  %.3f of the %s; s_memtime / s_memrealtime = %.0f MHz, %.1f cycles per MFMA and SIMD\n", MODE, DATA == 2 ? "e2m1 +-1" : (DATA ? "random  " : "constant"), what, best, flops / best / 1e9,
         flops / best / 1e9 / (DATA == 2 ? 10000.0 : 5000.0), DATA == 2 ? "10 PFLOP/s dense FP4 peak" : "5 PFLOP/s dense i8 peak", mhz, sc / n_wg / (32.0 * NTN * STEPS));
}

int main() {
  hipDeviceProp_t p;
  hipGetDeviceProperties(&p, 0);
  const int n_wg = p.multiProcessorCount * 4;  // four rounds of one workgroup per CU
  uint8_t *src;
  int *out;
  const size_t big = (size_t)2 << 30;
  hipMalloc(&src, big + (1 << 20));
  hipMemset(src, 1, big + (1 << 20));
  hipMalloc(&out, (size_t)n_wg * 512 * 4);
  printf("%s, %d CUs, %d workgroups of 512 threads, %d K-steps of 80 MFMAs per wave\n", p.name, p.multiProcessorCount, n_wg, STEPS);
  hipMemset(src, 1, big + (1 << 20));
  run<0, 0>(src, 0, out, n_wg, "MFMAs only");
  run<0, 1>(src, 0, out, n_wg, "MFMAs only");
  run<1, 0>(src, 0, out, n_wg, "+ fragment reads (19 ds_read_b128 per wave and K-step)");
  run<1, 1>(src, 0, out, n_wg, "+ fragment reads (19 ds_read_b128 per wave and K-step)");
  run<2, 1>(src, 0, out, n_wg, "+ one barrier per K-step");
  run<3, 0>(src, 0, out, n_wg, "+ LDS-DMA of the K-step's 72 KiB, source resident in L2");
  run<4, 0>(src, big / n_wg / 4096 * 4096, out, n_wg, "+ LDS-DMA, every workgroup streams its own 2 MiB (L2 misses)");
  {  // random bytes in the DMA source
    std::vector<uint32_t> h(((size_t)64 << 20) / 4);
    uint32_t x = 12345;
    for (auto &v : h) v = (x = x * 1664525u + 1013904223u);
    for (size_t o = 0; o < big; o += (size_t)64 << 20) hipMemcpy(src + o, h.data(), (size_t)64 << 20, hipMemcpyHostToDevice);
  }
  run<3, 1>(src, 0, out, n_wg, "+ LDS-DMA of the K-step's 72 KiB, source resident in L2");
  run<4, 1>(src, big / n_wg / 4096 * 4096, out, n_wg, "+ LDS-DMA, every workgroup streams its own 2 MiB (L2 misses)");
  // uneven M split between the loader waves and their SIMD partners (random bytes, L2-resident source)
  for (int rep = 0; rep < 2; ++rep) {
    run<3, 1, 8>(src, 0, out, n_wg, "L2-resident DMA, loaders 8 M tiles / partners 8 (as shipped)");
    run<3, 1, 6>(src, 0, out, n_wg, "L2-resident DMA, loaders 6 M tiles / partners 10");
    run<3, 1, 4>(src, 0, out, n_wg, "L2-resident DMA, loaders 4 M tiles / partners 12");
    // the same with 256 x 256 tiles (4 N tiles per wave: no spills in any split; the DMA still moves the 72 KiB of the wide tile)
    run<3, 1, 8, 4>(src, 0, out, n_wg, "256 x 256: loaders 8 / partners 8");
    run<3, 1, 6, 4>(src, 0, out, n_wg, "256 x 256: loaders 6 / partners 10");
    run<3, 1, 4, 4>(src, 0, out, n_wg, "256 x 256: loaders 4 / partners 12");
    run<3, 1, 2, 4>(src, 0, out, n_wg, "256 x 256: loaders 2 / partners 14");
  }
  {  // the Hamming search's operands: e2m1 nibbles 0x2 / 0xA in the DMA source, in LDS and in the registers
    std::vector<uint32_t> h(((size_t)64 << 20) / 4);
    for (size_t i = 0; i < h.size(); ++i) h[i] = nib((uint32_t)i);
    for (size_t o = 0; o < big; o += (size_t)64 << 20) hipMemcpy(src + o, h.data(), (size_t)64 << 20, hipMemcpyHostToDevice);
  }
  for (int rep = 0; rep < 2; ++rep) {
    run<0, 2>(src, 0, out, n_wg, "MFMAs only");
    run<1, 2>(src, 0, out, n_wg, "+ fragment reads");
    run<2, 2>(src, 0, out, n_wg, "+ one barrier per K-step");
    run<3, 2>(src, 0, out, n_wg, "+ LDS-DMA, source resident in L2 (256 x 320, 8 / 8)");
    run<4, 2>(src, big / n_wg / 4096 * 4096, out, n_wg, "+ LDS-DMA, every workgroup streams its own 2 MiB (L2 misses)");
    run<3, 2, 8, 4>(src, 0, out, n_wg, "256 x 256: loaders 8 / partners 8");
    run<3, 2, 6, 4>(src, 0, out, n_wg, "256 x 256: loaders 6 / partners 10");
    run<3, 2, 4, 4>(src, 0, out, n_wg, "256 x 256: loaders 4 / partners 12");
  }
  return 0;
}
