// Issue-rate microbenchmark for the integer instructions the k-mer kernel is made of.
// Prints cycles per wave-instruction per SIMD at 1, 2, 4 and 8 waves per SIMD.
//   hipcc --offload-arch=gfx950 -O3 tools/gpu_microbench.hip -o gpurun_out/microbench
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <vector>

#define ITERS 4096
#define REP 16

template <int OP>
__global__ void k(uint32_t *out, uint32_t seed, unsigned long long *cycles) {
  uint32_t a0 = threadIdx.x * 2654435761u + seed, a1 = a0 ^ 0x9e3779b9u, a2 = a0 + 77, a3 = a1 + 1234567;
  uint64_t b0 = ((uint64_t)a0 << 32) | a1, b1 = ((uint64_t)a2 << 32) | a3, b2 = b0 * 3 + 1, b3 = b1 * 5 + 7;
  unsigned long long t0 = __builtin_amdgcn_s_memtime();
  for (int i = 0; i < ITERS; ++i) {
#pragma unroll
    for (int r = 0; r < REP / 4; ++r) {
      if (OP == 0) {  // v_add_u32
        asm volatile("v_add_u32 %0, %0, %4\n v_add_u32 %1, %1, %4\n v_add_u32 %2, %2, %4\n v_add_u32 %3, %3, %4"
                     : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3) : "v"(seed));
      } else if (OP == 1) {  // v_mul_lo_u32
        asm volatile("v_mul_lo_u32 %0, %0, %4\n v_mul_lo_u32 %1, %1, %4\n v_mul_lo_u32 %2, %2, %4\n v_mul_lo_u32 %3, %3, %4"
                     : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3) : "v"(seed));
      } else if (OP == 2) {  // v_mad_u64_u32
        asm volatile("v_mad_u64_u32 %0, vcc, %4, %5, %0\n v_mad_u64_u32 %1, vcc, %4, %5, %1\n"
                     "v_mad_u64_u32 %2, vcc, %4, %5, %2\n v_mad_u64_u32 %3, vcc, %4, %5, %3"
                     : "+v"(b0), "+v"(b1), "+v"(b2), "+v"(b3) : "v"(a0), "v"(seed) : "vcc");
      } else if (OP == 3) {  // v_lshl_add_u64
        asm volatile("v_lshl_add_u64 %0, %0, 0, %4\n v_lshl_add_u64 %1, %1, 0, %4\n"
                     "v_lshl_add_u64 %2, %2, 0, %4\n v_lshl_add_u64 %3, %3, 0, %4"
                     : "+v"(b0), "+v"(b1), "+v"(b2), "+v"(b3) : "v"(b1));
      } else if (OP == 4) {  // v_perm_b32
        asm volatile("v_perm_b32 %0, %0, %4, %5\n v_perm_b32 %1, %1, %4, %5\n v_perm_b32 %2, %2, %4, %5\n v_perm_b32 %3, %3, %4, %5"
                     : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3) : "v"(seed), "v"(0x07020500u));
      } else if (OP == 5) {  // v_mov_b32
        asm volatile("v_mov_b32 %0, %4\n v_mov_b32 %1, %4\n v_mov_b32 %2, %4\n v_mov_b32 %3, %4"
                     : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3) : "v"(seed));
      } else if (OP == 6) {  // v_xor_b32
        asm volatile("v_xor_b32 %0, %0, %4\n v_xor_b32 %1, %1, %4\n v_xor_b32 %2, %2, %4\n v_xor_b32 %3, %3, %4"
                     : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3) : "v"(seed));
      } else if (OP == 7) {  // v_mul_hi_u32
        asm volatile("v_mul_hi_u32 %0, %0, %4\n v_mul_hi_u32 %1, %1, %4\n v_mul_hi_u32 %2, %2, %4\n v_mul_hi_u32 %3, %3, %4"
                     : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3) : "v"(seed));
      } else if (OP == 8) {  // v_cmp_lt_u64 + v_cndmask
        asm volatile("v_cmp_lt_u64 vcc, %4, %5\n v_cndmask_b32 %0, %0, %1, vcc\n v_cmp_lt_u64 vcc, %5, %4\n v_cndmask_b32 %2, %2, %3, vcc"
                     : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3) : "v"(b0), "v"(b1) : "vcc");
      } else if (OP == 9) {  // v_lshrrev_b64
        asm volatile("v_lshrrev_b64 %0, 3, %0\n v_lshrrev_b64 %1, 3, %1\n v_lshrrev_b64 %2, 3, %2\n v_lshrrev_b64 %3, 3, %3"
                     : "+v"(b0), "+v"(b1), "+v"(b2), "+v"(b3));
      } else if (OP == 10) {  // v_mul_u32_u24
        asm volatile("v_mul_u32_u24 %0, %0, %4\n v_mul_u32_u24 %1, %1, %4\n v_mul_u32_u24 %2, %2, %4\n v_mul_u32_u24 %3, %3, %4"
                     : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3) : "v"(seed));
      } else if (OP == 12) {  // v_add_co_u32 + v_addc_co_u32 pairs
        asm volatile("v_add_co_u32_e32 %0, vcc, %0, %4\n v_addc_co_u32_e32 %1, vcc, %1, %4, vcc\n"
                     "v_add_co_u32_e32 %2, vcc, %2, %4\n v_addc_co_u32_e32 %3, vcc, %3, %4, vcc"
                     : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3) : "v"(seed) : "vcc");
      } else if (OP == 13) {  // v_and_b32 with a 32-bit literal
        asm volatile("v_and_b32 %0, 0xdfdfdfdf, %0\n v_and_b32 %1, 0xdfdfdfdf, %1\n v_and_b32 %2, 0xdfdfdfdf, %2\n v_and_b32 %3, 0xdfdfdfdf, %3"
                     : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3));
      } else if (OP == 14) {  // v_bitop3_b32
        asm volatile("v_bitop3_b32 %0, %0, %4, %1 bitop3:0x96\n v_bitop3_b32 %1, %1, %4, %2 bitop3:0x96\n"
                     "v_bitop3_b32 %2, %2, %4, %3 bitop3:0x96\n v_bitop3_b32 %3, %3, %4, %0 bitop3:0x96"
                     : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3) : "v"(seed));
      } else if (OP == 15) {  // v_lshrrev_b32
        asm volatile("v_lshrrev_b32 %0, 1, %0\n v_lshrrev_b32 %1, 1, %1\n v_lshrrev_b32 %2, 1, %2\n v_lshrrev_b32 %3, 1, %3"
                     : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3));
      } else if (OP == 27) {  // v_bcnt_u32_b32 (popcount + add: the Hamming kernel's second instruction)
        asm volatile("v_bcnt_u32_b32 %0, %4, %0\n v_bcnt_u32_b32 %1, %4, %1\n v_bcnt_u32_b32 %2, %4, %2\n v_bcnt_u32_b32 %3, %4, %3"
                     : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3) : "v"(seed));
      } else if (OP == 28) {  // v_dot4_u32_u8 and v_dot2_i32_i16 style packed dot (k-mer classification / dist prepass)
        asm volatile("v_dot4_u32_u8 %0, %4, %5, %0\n v_dot4_u32_u8 %1, %4, %5, %1\n v_dot4_u32_u8 %2, %4, %5, %2\n v_dot4_u32_u8 %3, %4, %5, %3"
                     : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3) : "v"(seed), "v"(0x40100401u));
      } else if (OP == 16) {  // v_cndmask_b32 (VOP2, vcc)
        asm volatile("v_cndmask_b32 %0, %0, %4, vcc\n v_cndmask_b32 %1, %1, %4, vcc\n v_cndmask_b32 %2, %2, %4, vcc\n v_cndmask_b32 %3, %3, %4, vcc"
                     : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3) : "v"(seed) : "vcc");
      } else if (OP == 17) {  // v_lshl_or_b32 (VOP3)
        asm volatile("v_lshl_or_b32 %0, %0, 1, %4\n v_lshl_or_b32 %1, %1, 1, %4\n v_lshl_or_b32 %2, %2, 1, %4\n v_lshl_or_b32 %3, %3, 1, %4"
                     : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3) : "v"(seed));
      } else if (OP == 18) {  // v_add3_u32 (VOP3)
        asm volatile("v_add3_u32 %0, %0, %4, %1\n v_add3_u32 %1, %1, %4, %2\n v_add3_u32 %2, %2, %4, %3\n v_add3_u32 %3, %3, %4, %0"
                     : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3) : "v"(seed));
      } else if (OP == 19) {  // mixed: mad, mov, mad, xor  (mul128-like, dependent chain per accumulator)
        asm volatile("v_mad_u64_u32 %0, vcc, %4, %5, %0\n v_mov_b32 %2, %3\n v_mad_u64_u32 %1, vcc, %4, %5, %1\n v_xor_b32 %3, %2, %3"
                     : "+v"(b0), "+v"(b1), "+v"(a2), "+v"(a3) : "v"(a0), "v"(seed) : "vcc");
      } else if (OP == 20) {  // mixed: add, perm, add, alignbyte (2-cycle / 4-cycle alternation)
        asm volatile("v_add_u32 %0, %0, %4\n v_perm_b32 %1, %1, %4, %5\n v_add_u32 %2, %2, %4\n v_alignbyte_b32 %3, %3, %4, 1"
                     : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3) : "v"(seed), "v"(0x07020500u));
      } else if (OP == 21) {  // v_xor_b32 x4 but a single dependent chain
        asm volatile("v_xor_b32 %0, %0, %1\n v_xor_b32 %0, %0, %1\n v_xor_b32 %0, %0, %1\n v_xor_b32 %0, %0, %1"
                     : "+v"(a0) : "v"(seed));
      } else if (OP == 22) {  // v_mad_u64_u32 single dependent chain
        asm volatile("v_mad_u64_u32 %0, vcc, %1, %2, %0\n v_mad_u64_u32 %0, vcc, %1, %2, %0\n"
                     "v_mad_u64_u32 %0, vcc, %1, %2, %0\n v_mad_u64_u32 %0, vcc, %1, %2, %0"
                     : "+v"(b0) : "v"(a0), "v"(seed) : "vcc");
      } else if (OP == 23) {  // v_cndmask_b32_e64 with an SGPR-pair mask
        asm volatile("v_cndmask_b32_e64 %0, 0, 1, s[10:11]\n v_cndmask_b32_e64 %1, 0, 1, s[10:11]\n"
                     "v_cndmask_b32_e64 %2, 0, 1, s[10:11]\n v_cndmask_b32_e64 %3, 0, 1, s[10:11]"
                     : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3) : : "s10", "s11");
      } else if (OP == 24) {  // 1 compare + 3 v_cndmask on the same vcc (kernel's strand select pattern)
        asm volatile("v_cmp_lt_u64 vcc, %4, %5\n v_cndmask_b32 %0, %0, %1, vcc\n v_cndmask_b32 %2, %2, %3, vcc\n v_cndmask_b32 %1, %1, %2, vcc"
                     : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3) : "v"(b0), "v"(b1) : "vcc");
      } else if (OP == 25) {  // v_bfi_b32 select with a VGPR mask
        asm volatile("v_bfi_b32 %0, %4, %0, %1\n v_bfi_b32 %1, %4, %1, %2\n v_bfi_b32 %2, %4, %2, %3\n v_bfi_b32 %3, %4, %3, %0"
                     : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3) : "v"(seed));
      } else if (OP == 26) {  // v_cndmask vcc, vcc written once by a VALU compare before the loop body (4 per compare... 16 per loop trip)
        if (r == 0) asm volatile("v_cmp_lt_u32 vcc, %0, %1" : : "v"(a0), "v"(seed) : "vcc");
        asm volatile("v_cndmask_b32 %0, %0, %4, vcc\n v_cndmask_b32 %1, %1, %4, vcc\n v_cndmask_b32 %2, %2, %4, vcc\n v_cndmask_b32 %3, %3, %4, vcc"
                     : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3) : "v"(seed) : "vcc");
      } else if (OP == 11) {  // v_alignbyte_b32
        asm volatile("v_alignbyte_b32 %0, %0, %4, 1\n v_alignbyte_b32 %1, %1, %4, 2\n v_alignbyte_b32 %2, %2, %4, 3\n v_alignbyte_b32 %3, %3, %4, 1"
                     : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3) : "v"(seed));
      }
    }
  }
  unsigned long long t1 = __builtin_amdgcn_s_memtime();
  out[blockIdx.x * blockDim.x + threadIdx.x] = a0 ^ a1 ^ a2 ^ a3 ^ (uint32_t)(b0 ^ b1 ^ b2 ^ b3) ^ (uint32_t)((b0 ^ b1 ^ b2 ^ b3) >> 32);
  if (threadIdx.x == 0 && blockIdx.x == 0) *cycles = t1 - t0;
}

template <int OP>
void run(const char *name) {
  uint32_t *out;
  unsigned long long *cyc;
  hipMalloc(&out, 256 * 2048 * 4 * sizeof(uint32_t));
  hipMalloc(&cyc, 8);
  printf("%-18s", name);
  for (int wps : {1, 2, 4, 8}) {  // waves per SIMD: block of 256*wps threads, 1 block per CU
    int threads = 256 * wps > 1024 ? 1024 : 256 * wps;
    int blocks_per_cu = (256 * wps) / threads;
    hipEvent_t e0, e1;
    hipEventCreate(&e0), hipEventCreate(&e1);
    hipLaunchKernelGGL(k<OP>, dim3(256 * blocks_per_cu), dim3(threads), 0, 0, out, 12345u, cyc);
    hipDeviceSynchronize();
    hipEventRecord(e0);
    hipLaunchKernelGGL(k<OP>, dim3(256 * blocks_per_cu), dim3(threads), 0, 0, out, 12345u, cyc);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms;
    hipEventElapsedTime(&ms, e0, e1);
    unsigned long long c;
    hipMemcpy(&c, cyc, 8, hipMemcpyDeviceToHost);
    // s_memtime ticks at 100 MHz on gfx9; use wall time and an assumed clock instead:
    double n_instr_per_wave = (double)ITERS * REP;
    double instr_per_simd = n_instr_per_wave * wps;  // waves per SIMD
    double ns_per_instr = ms * 1e6 / instr_per_simd;
    printf("  wps%d: %6.2f ns/instr/SIMD (memtime %llu)", wps, ns_per_instr, c);
  }
  printf("\n");
  hipFree(out), hipFree(cyc);
}

int main() {
  printf("ns per wave-instruction per SIMD (at 2.4 GHz one cycle = 0.417 ns)\n");
  run<0>("v_add_u32");
  run<6>("v_xor_b32");
  run<5>("v_mov_b32");
  run<4>("v_perm_b32");
  run<11>("v_alignbyte_b32");
  run<3>("v_lshl_add_u64");
  run<9>("v_lshrrev_b64");
  run<8>("cmp_u64+cndmask");
  run<10>("v_mul_u32_u24");
  run<1>("v_mul_lo_u32");
  run<7>("v_mul_hi_u32");
  run<2>("v_mad_u64_u32");
  run<22>("mad_u64 1 chain");
  run<21>("v_xor 1 chain");
  run<12>("add_co+addc");
  run<13>("v_and literal");
  run<14>("v_bitop3_b32");
  run<15>("v_lshrrev_b32");
  run<16>("v_cndmask vcc");
  run<23>("v_cndmask_e64 sgpr");
  run<17>("v_lshl_or_b32");
  run<18>("v_add3_u32");
  run<24>("cmp64 + 3 cndmask");
  run<26>("cmp32 + 16 cndmask");
  run<25>("v_bfi_b32");
  run<27>("v_bcnt_u32_b32");
  run<28>("v_dot4_u32_u8");
  run<19>("mix mad/mov/mad/xor");
  run<20>("mix add/perm/add/align");
  return 0;
}
