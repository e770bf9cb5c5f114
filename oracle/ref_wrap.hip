// Build recipe only: compiles the reference's own kernel source, in place, for
// gfx950.  No reference text is copied into this repository.
#include <hip/hip_runtime.h>
#define static /* nvcc tolerates `extern "C" __device__ static inline`; clang does not */
#include HG_REF_CU
#undef static
