/*
 * hg_oracle_avx2.c -- encode_hash_hd_avx2 (src/hd.rs:14-92) with the REAL AVX2 intrinsics.
 *
 * TEST INFRASTRUCTURE ONLY (see hg_oracle.h).  The production sketches of the reference are written by the AVX2
 * routine, whose output order is a permutation of the scalar routine's (src/hd.rs:94-112).  hg_oracle.c states
 * that permutation in closed form (ORC_LAYOUT_AVX2) and also emulates the intrinsic sequence in scalar C; this
 * file removes the emulation from the chain of trust: it follows src/hd.rs:14-92 statement by statement with
 * <immintrin.h>, so the x86-64 CPU itself defines what the shuffles / hadds / permutes do.  Built with
 * `gcc -mavx2` into libhg_oracle_avx2.so (oracle/Makefile); tests/test_oracle.py compares it with orc_encode_hv.
 * The only restated piece left underneath is WyRng (orc_wyrng_next).
 */
#include <immintrin.h>
#include <stdlib.h>
#include <string.h>

#include "hg_oracle.h"

void orc_encode_hv_avx2_intrinsics(const uint64_t *hashes, size_t n, size_t hv_d, int16_t *hv) {
  const __m256i one = _mm256_set1_epi16(1);                      /* hd.rs:17 */
  const __m256i zero = _mm256_setzero_si256();                   /* hd.rs:18 */
  const __m256i shuffle_mask = _mm256_set_epi8(                  /* hd.rs:19-22 */
      15, 14, 7, 6, 13, 12, 5, 4, 11, 10, 3, 2, 9, 8, 1, 0, 15, 14, 7, 6, 13, 12, 5, 4, 11, 10, 3, 2, 9, 8, 1, 0);
  uint64_t rng[4], rnd[4];                                       /* hd.rs:24-26 */
  for (size_t d = 0; d < hv_d; ++d) hv[d] = (int16_t)(-(int16_t)n); /* hd.rs:29: -(num_seed as i16) */

  const size_t tail = n % 4;                                     /* hd.rs:31-34 */
  const size_t n4 = n + (tail == 0 ? 0 : 4 - tail);
  const size_t batches = n4 / 4, chunks = hv_d / 64;
  uint64_t *seeds = (uint64_t *)calloc(n4 ? n4 : 1, sizeof(uint64_t)); /* hd.rs:36-38: resize(.., 0) */
  if (!seeds) return;
  if (n) memcpy(seeds, hashes, n * sizeof(uint64_t));

  for (size_t b = 0; b < batches; ++b) {                         /* hd.rs:41 */
    for (int j = 0; j < 4; ++j) rng[j] = seeds[b * 4 + j];       /* hd.rs:43-45: seed_from_u64 = state */
    for (size_t i = 0; i < chunks; ++i) {                        /* hd.rs:48 */
      for (int j = 0; j < 4; ++j) rnd[j] = orc_wyrng_next(&rng[j]); /* hd.rs:50-52 */
      if (b == batches - 1 && tail > 0)                          /* hd.rs:54-58 */
        for (size_t j = tail; j < 4; ++j) rnd[j] = 0;
      const __m256i v = _mm256_shuffle_epi8(                     /* hd.rs:61-69 */
          _mm256_set_epi64x((long long)rnd[0], (long long)rnd[1], (long long)rnd[2], (long long)rnd[3]), shuffle_mask);
      for (int k = 0; k < 16; ++k) {                             /* hd.rs:71 */
        const __m256i bits = _mm256_and_si256(_mm256_srl_epi16(v, _mm_set1_epi64x(k)), one); /* hd.rs:72-76 */
        __m256i h = _mm256_hadd_epi16(bits, zero);               /* hd.rs:78 */
        h = _mm256_permute4x64_epi64(h, 0xD8);                   /* hd.rs:79 */
        h = _mm256_shuffle_epi8(h, shuffle_mask);                /* hd.rs:80 */
        h = _mm256_hadd_epi16(h, zero);                          /* hd.rs:81 */
        h = _mm256_slli_epi16(h, 1);                             /* hd.rs:82 */
        int16_t *o = hv + i * 64 + (size_t)k * 4;                /* hd.rs:84-87, wrapping i16 adds */
        o[0] = (int16_t)(o[0] + (int16_t)_mm256_extract_epi16(h, 0));
        o[1] = (int16_t)(o[1] + (int16_t)_mm256_extract_epi16(h, 1));
        o[2] = (int16_t)(o[2] + (int16_t)_mm256_extract_epi16(h, 2));
        o[3] = (int16_t)(o[3] + (int16_t)_mm256_extract_epi16(h, 3));
      }
    }
  }
  free(seeds);
}
