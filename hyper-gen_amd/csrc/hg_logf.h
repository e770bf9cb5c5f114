// hg_logf.h -- the natural logarithm the reference's ANI formula calls, bit for bit.
//
// `ani = 1.0 + (2.0 / (1.0 / jaccard + 1.0)).ln() / k` (src/dist.rs:154): Rust's f32::ln is the C library's logf, and on
// every glibc since 2.27 that is the table-driven routine of sysdeps/ieee754/flt-32/e_logf.c (ARM optimized-routines:
// 16 subintervals, log(x) = log1p(z / c - 1) + log(c) + k ln 2, a cubic in double).  It is NOT correctly rounded
// (0.82 ulp), so a "better" logf gives different last bits -- and with them a different third decimal in the TSV for a
// pair next to a rounding boundary, a different side of `ani >= ani_th`, a different order of near-ties.  The device
// therefore evaluates glibc's own algorithm: same table, same polynomial, same double arithmetic, in the form an x86-64
// host with FMA runs it (the ifunc picks __logf_fma there -- e_logf.c compiled with -mfma -mavx2, where the compiler
// fuses each of the source's five multiply-adds; the fusions below are read off its disassembly in glibc 2.35:
// r = fma(z, invc, -1), y0 = fma(k, Ln2, logc), y = fma(A1, r, A2), y = fma(A0, r2, y), y = fma(y, r2, r + y0)).
// A host without FMA runs the unfused form (__logf_sse2).  The two forms, and the host's logf of this image (glibc 2.35),
// were compared on all 2^32 inputs: they return the same float everywhere (their double results differ in the last bit
// for most inputs, never across a float rounding boundary) -- "glibc's logf" is one function of x.
// tests/test_gpu_ani_exact.py compares this function with the host's logf on every float in (0, 1].
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace hg_logf_detail {
// {1 / c, log(c)} for the 16 subintervals of [0x1.66p-1, 0x1.66p0) -- the values of glibc's __logf_data.tab
__device__ static const double TAB[32] = {
    0x1.661ec79f8f3bep+0, -0x1.57bf7808caadep-2, 0x1.571ed4aaf883dp+0, -0x1.2bef0a7c06ddbp-2,
    0x1.49539f0f010bp+0,  -0x1.01eae7f513a67p-2, 0x1.3c995b0b80385p+0, -0x1.b31d8a68224e9p-3,
    0x1.30d190c8864a5p+0, -0x1.6574f0ac07758p-3, 0x1.25e227b0b8eap+0,  -0x1.1aa2bc79c81p-3,
    0x1.1bb4a4a1a343fp+0, -0x1.a4e76ce8c0e5ep-4, 0x1.12358f08ae5bap+0, -0x1.1973c5a611cccp-4,
    0x1.0953f419900a7p+0, -0x1.252f438e10c1ep-5, 0x1p+0,               0x0p+0,
    0x1.e608cfd9a47acp-1, 0x1.aa5aa5df25984p-5,  0x1.ca4b31f026aap-1,  0x1.c5e53aa362eb4p-4,
    0x1.b2036576afce6p-1, 0x1.526e57720db08p-3,  0x1.9c2d163a1aa2dp-1, 0x1.bc2860d22477p-3,
    0x1.886e6037841edp-1, 0x1.1058bc8a07ee1p-2,  0x1.767dcf5534862p-1, 0x1.4043057b6ee09p-2};
}  // namespace hg_logf_detail

__device__ __forceinline__ float hg_logf(float x) {
  constexpr double LN2 = 0x1.62e42fefa39efp-1;
  constexpr double A0 = -0x1.00ea348b88334p-2, A1 = 0x1.5575b0be00b6ap-2, A2 = -0x1.ffffef20a4123p-2;
  uint32_t ix = __float_as_uint(x);
  if (ix == 0x3f800000u) return 0.0f;  // log(1) is exactly +0
  if (ix - 0x00800000u >= 0x7f800000u - 0x00800000u) {
    // x < 0x1p-126, or inf, or nan
    if (ix * 2u == 0u) return -__builtin_inff();       // log(+-0) = -inf
    if (ix == 0x7f800000u) return x;                    // log(inf) = inf
    if ((ix & 0x80000000u) || ix * 2u >= 0xff000000u) return __builtin_nanf("");  // x < 0 or nan
    ix = __float_as_uint(x * 0x1p23f);                  // subnormal: normalise
    ix -= 23u << 23;
  }
  // x = 2^k z, z in [0x1.66p-1, 0x1.66p0), exact; subinterval i holds z, c is near its centre
  const uint32_t tmp = ix - 0x3f330000u;
  const uint32_t i = (tmp >> 19) & 15u;
  const int32_t k = (int32_t)tmp >> 23;
  const uint32_t iz = ix - (tmp & 0xff800000u);
  const double invc = hg_logf_detail::TAB[2 * i], logc = hg_logf_detail::TAB[2 * i + 1];
  const double z = (double)__uint_as_float(iz);
  const double r = __builtin_fma(z, invc, -1.0);
  const double y0 = __builtin_fma((double)k, LN2, logc);
  const double r2 = r * r;
  double y = __builtin_fma(A1, r, A2);
  y = __builtin_fma(A0, r2, y);
  y = __builtin_fma(y, r2, r + y0);
  return (float)y;  // round to nearest even, like cvtsd2ss
}
