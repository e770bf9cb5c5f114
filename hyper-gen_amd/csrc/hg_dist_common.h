// hg_dist_common.h -- what every part of the all-pairs ANI path shares (private to hg_dist_kernels.hip).
#pragma once
#include <type_traits>
#include <utility>

#include "hg_internal.h"
#include "hg_logf.h"

namespace {

typedef _Float16 half8 __attribute__((ext_vector_type(8)));
typedef float float4v __attribute__((ext_vector_type(4)));
typedef int int4v __attribute__((ext_vector_type(4)));
typedef int int8v __attribute__((ext_vector_type(8)));
template <int... Js, class F>
__device__ __forceinline__ void dist_static_for(std::integer_sequence<int, Js...>, F &&f) {
  (f(std::integral_constant<int, Js>{}), ...);
}

// ---- ANI epilogue (src/dist.rs:153-160) -----------------------------------------------------
__device__ __forceinline__ float ani_from_dot(int32_t dot, int32_t nr, int32_t nq, float kf) {
  const int32_t den = (int32_t)((uint32_t)nr + (uint32_t)nq - (uint32_t)dot);  // i32 wrapping
  const float jaccard = (float)dot / (float)den;
  const float inner = 1.0f / jaccard + 1.0f;
  const float x = 2.0f / inner;
  float ani = 1.0f + hg_logf(x) / kf;  // (glibc's logf, bit for bit: hg_logf.h)
  if (ani != ani) return 0.0f;  // is_nan -> 0
  ani = fminf(ani, 1.0f);
  ani = fmaxf(ani, 0.0f);
  return ani * 100.0f;
}

}  // namespace
