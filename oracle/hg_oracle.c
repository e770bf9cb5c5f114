/*
 * hg_oracle.c -- plain-C CPU restatement of the HyperGen sketch + ANI hot path.
 *
 * TEST INFRASTRUCTURE ONLY (see hg_oracle.h).  Written from the behaviour of the
 * reference (file:line cited per function); no reference source is copied.
 *
 * Pinning (details in oracle/README.md):
 *   - t1ha2: upstream self-check values + the reference's own CUDA restatement
 *     (src/cuda_kernel.cu compiled by hipcc into oracle/_ref, run on the GPU box)
 *     + the G1 vectors of SURVEY.md 8c (tests/golden/g1_*.json).
 *   - WyRng: upstream README known answer.
 *   - AVX2 HV layout: closed form checked against an emulation of the intrinsic
 *     sequence of src/hd.rs:14-92.
 *   - BitPacker8x / bincode layouts: restated from the published crate
 *     algorithms; NO reference-produced .sketch exists here => parity unpinned.
 */
#include "hg_oracle.h"

#include <math.h>
#ifdef _OPENMP
#include <omp.h>
#endif
#include <stdlib.h>
#include <string.h>

/* ------------------------------------------------------------------------- */
/* t1ha2                                                                      */
/* ------------------------------------------------------------------------- */

/* primes: src/cuda_kernel.cu:71-77 (same constants as upstream t1ha) */
static const uint64_t P0 = 0xEC99BF0D8372CAABull;
static const uint64_t P1 = 0x82434FE90EDCEF39ull;
static const uint64_t P2 = 0xD4F06DB99D67BE4Bull;
static const uint64_t P3 = 0xBD9CACC22C6E9571ull;
static const uint64_t P4 = 0x9C06FAF4D023E3ABull;
static const uint64_t P5 = 0xC060724A8424F345ull;
static const uint64_t P6 = 0xCB5AF53AE3AAAC31ull;

static inline uint64_t rot64(uint64_t v, unsigned s) {
  return (v >> s) | (v << (64 - s));
}

/* little-endian load of `n` (1..8) bytes: src/cuda_kernel.cu:155-194 */
static inline uint64_t load_le(const uint8_t *p, unsigned n) {
  uint64_t r = 0;
  for (unsigned i = 0; i < n; i++) r |= (uint64_t)p[i] << (8 * i);
  return r;
}

/* src/cuda_kernel.cu:136-141 */
static inline void mixup64(uint64_t *a, uint64_t *b, uint64_t v, uint64_t prime) {
  unsigned __int128 m = (unsigned __int128)(*b + v) * prime;
  *a ^= (uint64_t)m;
  *b += (uint64_t)(m >> 64);
}

/* src/cuda_kernel.cu:143-153 */
static inline uint64_t final64(uint64_t a, uint64_t b) {
  uint64_t x = (a + rot64(b, 41)) * P0;
  uint64_t y = (rot64(a, 23) + b) * P6;
  unsigned __int128 m = (unsigned __int128)(x ^ y) * P5;
  return (uint64_t)m ^ (uint64_t)(m >> 64);
}

uint64_t orc_t1ha2_atonce(const uint8_t *data, size_t len, uint64_t seed) {
  uint64_t a = seed, b = (uint64_t)len; /* src/cuda_kernel.cu:200-203 */

  if (len > 32) {
    /* published t1ha2 long-input path (absent from src/cuda_kernel.cu, present
     * in the t1ha crate the CPU path calls at src/sketch.rs:90): two extra
     * lanes c,d, 32 bytes per round, then squash into a,b. */
    uint64_t c = rot64((uint64_t)len, 23) + ~seed;
    uint64_t d = ~(uint64_t)len + rot64(seed, 19);
    const uint8_t *detent = data + len - 31;
    do {
      uint64_t w0 = load_le(data, 8), w1 = load_le(data + 8, 8);
      uint64_t w2 = load_le(data + 16, 8), w3 = load_le(data + 24, 8);
      data += 32;
      uint64_t d02 = w0 + rot64(w2 + d, 56);
      uint64_t c13 = w1 + rot64(w3 + c, 19);
      d ^= b + rot64(w1, 38);
      c ^= a + rot64(w0, 57);
      b ^= P6 * (c13 + w2);
      a ^= P5 * (d02 + w3);
    } while (data < detent);
    a ^= P6 * (c + rot64(d, 23));
    b ^= P5 * (rot64(c, 19) + d);
    len &= 31;
  }

  /* tail switch: src/cuda_kernel.cu:207-245 */
  size_t rem = len;
  if (rem > 24) {
    mixup64(&a, &b, load_le(data, 8), P4);
    data += 8, rem -= 8;
  }
  if (rem > 16) {
    mixup64(&b, &a, load_le(data, 8), P3);
    data += 8, rem -= 8;
  }
  if (rem > 8) {
    mixup64(&a, &b, load_le(data, 8), P2);
    data += 8, rem -= 8;
  }
  if (rem > 0) mixup64(&b, &a, load_le(data, (unsigned)rem), P1);
  return final64(a, b);
}

/* ------------------------------------------------------------------------- */
/* WyRng                                                                      */
/* ------------------------------------------------------------------------- */

uint64_t orc_wyrng_next(uint64_t *state) {
  /* wyhash crate v1 wyrng: s += P0; mum(s ^ P1, s) */
  *state += 0xa0761d6478bd642full;
  unsigned __int128 m =
      (unsigned __int128)(*state ^ 0xe7037ed1a0b428dbull) * (*state);
  return (uint64_t)(m >> 64) ^ (uint64_t)m;
}

/* ------------------------------------------------------------------------- */
/* k-mer sampling                                                             */
/* ------------------------------------------------------------------------- */

/* 0 => not a base (run breaks); else the upper-case base */
static inline uint8_t norm_base(uint8_t c, int norm_mode) {
  switch (c) {
    case 'A': case 'a': return 'A';
    case 'C': case 'c': return 'C';
    case 'G': case 'g': return 'G';
    case 'T': case 't': return 'T';
    case 'U': case 'u': return norm_mode == ORC_NORM_U2T ? 'T' : 0;
    default: return 0;
  }
}

static inline uint8_t comp_base(uint8_t c) {
  switch (c) {
    case 'A': return 'T';
    case 'C': return 'G';
    case 'G': return 'C';
    case 'T': return 'A';
    default: return c; /* src/cuda_kernel.cu:293-296: c_rev = c_fwd */
  }
}

/* fwd / rc: scratch of n_bps bytes each */
static size_t kmer_sample_core(const uint8_t *seq, size_t n_bps, unsigned ksize,
                               uint64_t threshold, uint64_t seed, int canonical,
                               int norm_mode, uint64_t *out, size_t cap,
                               uint8_t *fwd, uint8_t *rc) {
  /* like needletail (src/sketch.rs:84-87): a normalised copy and its reverse
   * complement for the whole buffer, windows are slices of the two. */
  for (size_t i = 0; i < n_bps; i++) {
    uint8_t b = norm_base(seq[i], norm_mode);
    fwd[i] = b ? b : 'N';
    rc[n_bps - 1 - i] = b ? comp_base(b) : 'N';
  }
  size_t n_hit = 0, run = 0;
  for (size_t e = 0; e < n_bps; e++) { /* e = last base of the window */
    run = (fwd[e] != 'N') ? run + 1 : 0;
    if (run < ksize) continue;
    size_t s = e + 1 - ksize;
    const uint8_t *kf = fwd + s;
    const uint8_t *kr = rc + (n_bps - 1 - e);
    const uint8_t *km = kf;
    /* src/cuda_kernel.cu:306-311 / needletail canonical: smaller of the two */
    if (canonical && memcmp(kr, kf, ksize) < 0) km = kr;
    uint64_t h = orc_t1ha2_atonce(km, ksize, seed);
    if (h < threshold) { /* strict: src/sketch.rs:92, src/cuda_kernel.cu:316 */
      if (n_hit < cap) out[n_hit] = h;
      n_hit++;
    }
  }
  return n_hit;
}

size_t orc_kmer_hash_sample(const uint8_t *seq, size_t n_bps, unsigned ksize,
                            uint64_t threshold, uint64_t seed, int canonical,
                            int norm_mode, uint64_t *out, size_t cap) {
  if (ksize == 0 || n_bps < ksize) return 0;
  uint8_t *fwd = (uint8_t *)malloc(n_bps);
  uint8_t *rc = (uint8_t *)malloc(n_bps);
  if (!fwd || !rc) {
    free(fwd), free(rc);
    return 0;
  }
  size_t n = kmer_sample_core(seq, n_bps, ksize, threshold, seed, canonical, norm_mode, out, cap, fwd, rc);
  free(fwd), free(rc);
  return n;
}

static int cmp_u64(const void *a, const void *b) {
  uint64_t x = *(const uint64_t *)a, y = *(const uint64_t *)b;
  return (x > y) - (x < y);
}

size_t orc_sort_unique_u64(uint64_t *v, size_t n) {
  if (n == 0) return 0;
  qsort(v, n, sizeof(uint64_t), cmp_u64);
  size_t m = 1;
  for (size_t i = 1; i < n; i++)
    if (v[i] != v[m - 1]) v[m++] = v[i];
  return m;
}

size_t orc_read_merge_seq(const uint8_t *text, size_t n_text, uint8_t *out) {
  /* src/fastx_reader.rs:14-26 */
  size_t o = 0, i = 0;
  while (i < n_text) {
    size_t j = i;
    while (j < n_text && text[j] != '\n') j++;
    size_t end = j; /* line = [i, end), without the '\n' */
    if (text[i] == '>') {
      out[o++] = 'N';
    } else {
      size_t e = end;
      if (e > i && text[e - 1] == '\r') e--; /* :19-21, with or without '\n' */
      memcpy(out + o, text + i, e - i);
      o += e - i;
    }
    i = (j < n_text) ? j + 1 : j;
  }
  return o;
}

/* ------------------------------------------------------------------------- */
/* HV encode                                                                  */
/* ------------------------------------------------------------------------- */

void orc_encode_hv(const uint64_t *hashes, size_t n, size_t hv_d, int layout,
                   int16_t *hv) {
  size_t n_chunk = hv_d / 64; /* src/hd.rs:34,102: floor */
  uint16_t init = (uint16_t)(0u - (uint16_t)n); /* -(n as i16): src/hd.rs:29,97 */
  for (size_t d = 0; d < hv_d; d++) hv[d] = (int16_t)init;
  for (size_t h = 0; h < n; h++) {
    uint64_t st = hashes[h]; /* seed_from_u64(h): state = h */
    for (size_t i = 0; i < n_chunk; i++) {
      uint64_t w = orc_wyrng_next(&st);
      for (unsigned j = 0; j < 64; j++) {
        /* scalar: src/hd.rs:105-107.  avx2: bit j of the word lands at
         * 4*(j%16) + j/16 inside the chunk (src/hd.rs:19-22,61-87). */
        size_t pos = (layout == ORC_LAYOUT_AVX2) ? (4 * (j & 15) + (j >> 4)) : j;
        uint16_t *p = (uint16_t *)&hv[i * 64 + pos];
        *p = (uint16_t)(*p + (uint16_t)(((w >> j) & 1) << 1));
      }
    }
  }
}

/* --- emulation of the AVX2 intrinsic sequence, 16 x u16 "registers" --- */
typedef struct { uint16_t e[16]; } v256;

static v256 emu_shuffle_epi8(v256 a, const uint8_t mask[32]) {
  /* _mm256_shuffle_epi8: per 128-bit lane byte gather */
  uint8_t in[32], o[32];
  memcpy(in, a.e, 32);
  for (int lane = 0; lane < 2; lane++)
    for (int i = 0; i < 16; i++) {
      uint8_t m = mask[lane * 16 + i];
      o[lane * 16 + i] = (m & 0x80) ? 0 : in[lane * 16 + (m & 15)];
    }
  v256 r;
  memcpy(r.e, o, 32);
  return r;
}
static v256 emu_hadd_epi16(v256 a, v256 b) {
  v256 r;
  for (int lane = 0; lane < 2; lane++) {
    for (int i = 0; i < 4; i++) {
      r.e[lane * 8 + i] = (uint16_t)(a.e[lane * 8 + 2 * i] + a.e[lane * 8 + 2 * i + 1]);
      r.e[lane * 8 + 4 + i] = (uint16_t)(b.e[lane * 8 + 2 * i] + b.e[lane * 8 + 2 * i + 1]);
    }
  }
  return r;
}
static v256 emu_permute4x64(v256 a, unsigned imm) {
  v256 r;
  for (int q = 0; q < 4; q++) {
    unsigned src = (imm >> (2 * q)) & 3;
    memcpy(&r.e[4 * q], &a.e[4 * src], 8);
  }
  return r;
}

void orc_encode_hv_avx2_emulated(const uint64_t *hashes, size_t n, size_t hv_d,
                                 int16_t *hv) {
  /* _mm256_set_epi8 lists bytes 31..0: src/hd.rs:19-22 */
  static const uint8_t hi2lo[32] = {15, 14, 7, 6, 13, 12, 5, 4, 11, 10, 3, 2, 9, 8, 1, 0,
                                    15, 14, 7, 6, 13, 12, 5, 4, 11, 10, 3, 2, 9, 8, 1, 0};
  uint8_t mask[32];
  for (int i = 0; i < 32; i++) mask[i] = hi2lo[31 - i];
  v256 zero;
  memset(&zero, 0, sizeof zero);

  size_t n_chunk = hv_d / 64;
  uint16_t init = (uint16_t)(0u - (uint16_t)n);
  for (size_t d = 0; d < hv_d; d++) hv[d] = (int16_t)init;
  size_t tail = n % 4, n4 = n + (tail ? 4 - tail : 0), nb = n4 / 4;
  for (size_t b = 0; b < nb; b++) {
    uint64_t st[4];
    for (int j = 0; j < 4; j++) st[j] = (b * 4 + j < n) ? hashes[b * 4 + j] : 0; /* :36-38 */
    for (size_t i = 0; i < n_chunk; i++) {
      uint64_t rnd[4];
      for (int j = 0; j < 4; j++) rnd[j] = orc_wyrng_next(&st[j]);
      if (b == nb - 1 && tail > 0)
        for (size_t j = tail; j < 4; j++) rnd[j] = 0; /* :54-58 */
      /* _mm256_set_epi64x(r0,r1,r2,r3): r3 is the lowest qword */
      v256 x;
      uint64_t q[4] = {rnd[3], rnd[2], rnd[1], rnd[0]};
      memcpy(x.e, q, 32);
      x = emu_shuffle_epi8(x, mask);
      for (unsigned k = 0; k < 16; k++) {
        v256 s;
        for (int t = 0; t < 16; t++) s.e[t] = (uint16_t)((x.e[t] >> k) & 1);
        v256 h = emu_hadd_epi16(s, zero);
        h = emu_permute4x64(h, 0xD8);
        h = emu_shuffle_epi8(h, mask);
        h = emu_hadd_epi16(h, zero);
        for (int t = 0; t < 16; t++) h.e[t] = (uint16_t)(h.e[t] << 1);
        for (int m = 0; m < 4; m++) {
          uint16_t *p = (uint16_t *)&hv[i * 64 + k * 4 + m];
          *p = (uint16_t)(*p + h.e[m]);
        }
      }
    }
  }
}

int32_t orc_hv_norm2(const int16_t *hv, size_t hv_d) {
  uint32_t s = 0; /* i32 wrapping: Cargo.toml:63-65 */
  for (size_t d = 0; d < hv_d; d++) s += (uint32_t)((int32_t)hv[d] * (int32_t)hv[d]);
  return (int32_t)s;
}

/* ------------------------------------------------------------------------- */
/* bit packing                                                                */
/* ------------------------------------------------------------------------- */

unsigned orc_quant_bits(const int16_t *hv, size_t hv_d) {
  int16_t mn = hv[0], mx = hv[0];
  for (size_t d = 1; d < hv_d; d++) {
    if (hv[d] < mn) mn = hv[d];
    if (hv[d] > mx) mx = hv[d];
  }
  unsigned q = 6; /* src/hd.rs:123-136 */
  for (;;) {
    int16_t qmin = (int16_t)(-(1 << (q - 1)));
    int16_t qmax = (int16_t)((1 << (q - 1)) - 1);
    if (qmin <= mn && qmax >= mx) break;
    if (q == 16) break;
    q++;
  }
  return q;
}

/* BitPacker8x (crate bitpacking 0.9.2, `pack_unpack_with_bits!` over 8 x u32 lanes), restated from the
 * crate's DEFINITION of the format rather than from its accumulator loop (the product's hg_formats.cpp follows the
 * loop; two restatements of one reading would share a misreading):
 *   - a block is 256 values; value i belongs to lane l = i % 8 and is that lane's element r = i / 8;
 *   - lane l owns a bit stream of 32*q bits; element r starts at stream bit p = r*q (LSB first);
 *   - stream word w (32 bits) of lane l is the block's output u32 number 8*w + l, little endian.
 * The packer does not mask its input: an element wider than q bits spills into what follows it -- inside its first
 * word (a 32-bit left shift drops what passes bit 31) and, only if the element straddles a word boundary, as
 * `v >> (32 - p % 32)` into the next word.  That is what makes the reference's q = 16 path lossy (src/hd.rs:140-141:
 * `1 << 15` as i16 is -32768, `(i + offset) as u32` sign-extends). */
static void bp8x_scatter_block(const uint32_t *in, unsigned q, uint32_t *out) {
  memset(out, 0, (size_t)32 * q);
  for (unsigned i = 0; i < 256; i++) {
    const unsigned lane = i & 7, p = (i >> 3) * q, w0 = p >> 5, c = p & 31;
    out[8 * w0 + lane] |= in[i] << c;
    if (c + q > 32) out[8 * (w0 + 1) + lane] |= in[i] >> (32 - c); /* straddles: c > 0 here */
  }
}

/* element r of lane l = bits [r*q, r*q + q) of the lane's stream, read through a 64-bit window */
static void bp8x_gather_block(const uint32_t *in, unsigned q, uint32_t *out) {
  for (unsigned i = 0; i < 256; i++) {
    const unsigned lane = i & 7, p = (i >> 3) * q, w0 = p >> 5, c = p & 31;
    uint64_t win = in[8 * w0 + lane];
    if (w0 + 1 < q) win |= (uint64_t)in[8 * (w0 + 1) + lane] << 32;
    out[i] = (uint32_t)((win >> c) & ((1ull << q) - 1));
  }
}

/* bytes of a packed sketch: src/hd.rs:146 `vec![0u8; quant_bit * (hv_d >> 3)]` */
size_t orc_packed_bytes(size_t hv_d, unsigned q) { return (size_t)q * (hv_d >> 3); }

/* src/hd.rs:138-157.  Only whole 256-blocks are packed (:147 `hv_d / BLOCK_LEN`): the bytes behind them stay 0. */
void orc_pack_hv(const int16_t *hv, size_t hv_d, unsigned q, uint8_t *out) {
  const int16_t offset = (int16_t)(1u << (q - 1)); /* :140, i16: -32768 at q = 16 */
  uint32_t blk[256], packed[8 * 16];
  memset(out, 0, orc_packed_bytes(hv_d, q));
  for (size_t b = 0; b < hv_d / 256; b++) {
    for (unsigned i = 0; i < 256; i++)
      blk[i] = (uint32_t)(int32_t)(int16_t)((uint16_t)hv[b * 256 + i] + (uint16_t)offset); /* :141 wrapping i16, then `as u32` */
    bp8x_scatter_block(blk, q, packed);
    memcpy(out + (size_t)32 * q * b, packed, (size_t)32 * q);
  }
}

/* src/hd.rs:186-213.  Elements behind the last whole block decode from the zero-initialised u32 vector (:194):
 * `0 as i16 - offset`. */
void orc_unpack_hv(const uint8_t *packed, size_t hv_d, unsigned q, int16_t *hv) {
  const int16_t offset = (int16_t)(1u << (q - 1));
  uint32_t blk[256], in[8 * 16];
  for (size_t b = 0; b < hv_d / 256; b++) {
    memcpy(in, packed + (size_t)32 * q * b, (size_t)32 * q);
    bp8x_gather_block(in, q, blk);
    for (unsigned i = 0; i < 256; i++)
      hv[b * 256 + i] = (int16_t)((uint16_t)blk[i] - (uint16_t)offset); /* :206-212 */
  }
  for (size_t d = hv_d / 256 * 256; d < hv_d; d++) hv[d] = (int16_t)(0u - (uint16_t)offset);
}

/* The OTHER layout: what the reference writes / reads on a host WITHOUT AVX2 (src/hd.rs:158-166 and :213-231).  Restated bit
 * by bit in the reference's own loop order, in i16 arithmetic with Rust's release-mode semantics (wrapping; a shift amount
 * is taken modulo the type's width):
 *   pack  : (q*hv_d + 16) / 16 i16 words, zero-initialised (one word more than the bits need when q*hv_d is a multiple of 16);
 *           bit i of the stream = bit (i % q) of hv[i / q] -- the low q bits of the two's-complement VALUE, no offset added --
 *           stored at bit (i % 16) of word i / 16;
 *   unpack: the same bits OR-ed back, and at the end of every element `if v > (1 << (q-1)) { v - (1 << q) }` -- strictly
 *           greater, so the one value whose low q bits are exactly 100..0 (-2^(q-1)) comes back as +2^(q-1): the reference's
 *           naive round trip is not lossless for it.  q = 16: `1 << 15` is -32768 as i16 and `1 << 16` wraps to 1, so every
 *           element but -32768 comes back one lower.  Both are reproduced here (the layout is defined by what the reference
 *           does), and documented in include/hypergen.h. */
size_t orc_packed_words_naive(size_t hv_d, unsigned q) { return ((size_t)q * hv_d + 16) / 16; }
void orc_pack_hv_naive(const int16_t *hv, size_t hv_d, unsigned q, int16_t *out) {
  memset(out, 0, orc_packed_words_naive(hv_d, q) * sizeof(int16_t));
  for (size_t i = 0; i < (size_t)q * hv_d; i++) {
    const int16_t bit = (int16_t)((hv[i / q] >> (i % q)) & 1); /* arithmetic shift of the i16 value */
    out[i / 16] = (int16_t)((uint16_t)out[i / 16] | (uint16_t)((uint16_t)bit << (i % 16)));
  }
}
void orc_unpack_hv_naive(const int16_t *packed, size_t hv_d, unsigned q, int16_t *hv) {
  const int16_t half = (int16_t)(uint16_t)(1u << ((q - 1) & 15)); /* `1 << (q-1)` in i16 */
  const int16_t full = (int16_t)(uint16_t)(1u << (q & 15));       /* `1 << q` in i16: the amount wraps at 16 */
  memset(hv, 0, hv_d * sizeof(int16_t));
  for (size_t i = 0; i < (size_t)q * hv_d; i++) {
    const uint16_t bit = (uint16_t)((packed[i / 16] >> (i % 16)) & 1);
    hv[i / q] = (int16_t)((uint16_t)hv[i / q] | (uint16_t)(bit << (i % q)));
    if ((i + 1) % q == 0 && hv[i / q] > half) hv[i / q] = (int16_t)((uint16_t)hv[i / q] - (uint16_t)full);
  }
}

/* ------------------------------------------------------------------------- */
/* ANI                                                                        */
/* ------------------------------------------------------------------------- */

int32_t orc_hv_dot(const int16_t *r, const int16_t *q, size_t hv_d) {
  uint32_t s = 0;
  for (size_t d = 0; d < hv_d; d++) s += (uint32_t)((int32_t)r[d] * (int32_t)q[d]);
  return (int32_t)s;
}

float orc_ani_from_dot(int32_t dot, int32_t nr, int32_t nq, unsigned ksize) {
  /* src/dist.rs:153-160; the denominator is summed in (wrapping) i32 first */
  int32_t den = (int32_t)((uint32_t)nr + (uint32_t)nq - (uint32_t)dot);
  volatile float jaccard = (float)dot / (float)den;
  volatile float inner = 1.0f / jaccard + 1.0f;
  volatile float x = 2.0f / inner;
  float ani = 1.0f + logf(x) / (float)ksize;
  if (isnan(ani)) return 0.0f;
  ani = ani < 1.0f ? ani : 1.0f; /* .min(1.0) */
  ani = ani > 0.0f ? ani : 0.0f; /* .max(0.0) */
  return ani * 100.0f;
}

/* ---- logf --------------------------------------------------------------------
 * src/dist.rs:154 calls f32::ln = the C library's logf.  glibc >= 2.27 implements it as
 * sysdeps/ieee754/flt-32/e_logf.c (table of 16 {1/c, log c}, cubic in double; NOT correctly rounded), and on
 * x86-64 an ifunc picks the build of that file with -mfma where the CPU has FMA: there the compiler fuses the
 * five multiply-adds of the source.  Both forms are restated here from the published algorithm.  Compared
 * exhaustively in this image (glibc 2.35, a host with FMA): the host's logf, the fused form and the unfused form
 * return the same float for every one of the 2^32 inputs (orc_logf_sweep: 0 mismatches for either form; the
 * double results differ in their last bit for most inputs, never across a float rounding boundary) -- so "glibc's
 * logf" is ONE function of x, whatever the host CPU.  The device function hyper-gen_amd/csrc/hg_logf.h is the fused
 * form; tests/test_gpu_ani_exact.py compares it with the host's logf on every float in (0, 1].
 * The table values are glibc's __logf_data (the same bytes as in this image's libm.so.6 .rodata). */
static const double ORC_LOGF_TAB[32] = {
    0x1.661ec79f8f3bep+0, -0x1.57bf7808caadep-2, 0x1.571ed4aaf883dp+0, -0x1.2bef0a7c06ddbp-2,
    0x1.49539f0f010bp+0,  -0x1.01eae7f513a67p-2, 0x1.3c995b0b80385p+0, -0x1.b31d8a68224e9p-3,
    0x1.30d190c8864a5p+0, -0x1.6574f0ac07758p-3, 0x1.25e227b0b8eap+0,  -0x1.1aa2bc79c81p-3,
    0x1.1bb4a4a1a343fp+0, -0x1.a4e76ce8c0e5ep-4, 0x1.12358f08ae5bap+0, -0x1.1973c5a611cccp-4,
    0x1.0953f419900a7p+0, -0x1.252f438e10c1ep-5, 0x1p+0,               0x0p+0,
    0x1.e608cfd9a47acp-1, 0x1.aa5aa5df25984p-5,  0x1.ca4b31f026aap-1,  0x1.c5e53aa362eb4p-4,
    0x1.b2036576afce6p-1, 0x1.526e57720db08p-3,  0x1.9c2d163a1aa2dp-1, 0x1.bc2860d22477p-3,
    0x1.886e6037841edp-1, 0x1.1058bc8a07ee1p-2,  0x1.767dcf5534862p-1, 0x1.4043057b6ee09p-2};

float orc_logf_glibc(float x, int fused) {
  const double LN2 = 0x1.62e42fefa39efp-1;
  const double A0 = -0x1.00ea348b88334p-2, A1 = 0x1.5575b0be00b6ap-2, A2 = -0x1.ffffef20a4123p-2;
  uint32_t ix;
  memcpy(&ix, &x, 4);
  if (ix == 0x3f800000u) return 0.0f;
  if (ix - 0x00800000u >= 0x7f800000u - 0x00800000u) {
    if (ix * 2u == 0u) return -INFINITY;
    if (ix == 0x7f800000u) return x;
    if ((ix & 0x80000000u) || ix * 2u >= 0xff000000u) return NAN;
    volatile float xs = x * 0x1p23f; /* subnormal: normalise */
    float t = xs;
    memcpy(&ix, &t, 4);
    ix -= 23u << 23;
  }
  const uint32_t tmp = ix - 0x3f330000u;
  const uint32_t i = (tmp >> 19) & 15u;
  const int32_t k = (int32_t)tmp >> 23;
  const uint32_t iz = ix - (tmp & 0xff800000u);
  const double invc = ORC_LOGF_TAB[2 * i], logc = ORC_LOGF_TAB[2 * i + 1];
  float zf;
  memcpy(&zf, &iz, 4);
  const double z = (double)zf;
  if (fused) { /* __logf_fma: every multiply-add of the source is one fma */
    const double r = fma(z, invc, -1.0);
    const double y0 = fma((double)k, LN2, logc);
    const double r2 = r * r;
    double y = fma(A1, r, A2);
    y = fma(A0, r2, y);
    y = fma(y, r2, r + y0);
    return (float)y;
  }
  /* __logf_sse2: each product rounded on its own (volatile: no contraction whatever the compiler flags) */
  volatile double p;
  p = z * invc;
  const double r = p - 1.0;
  p = (double)k * LN2;
  const double y0 = logc + p;
  p = r * r;
  const double r2 = p;
  p = A1 * r;
  double y = p + A2;
  p = A0 * r2;
  y = p + y;
  p = y * r2;
  y = p + (y0 + r);
  return (float)y;
}

/* bit patterns in [first_bits, first_bits + n) on which the host's logf differs from the restatement (NaN == NaN);
 * *first_bad = the first of them */
uint64_t orc_logf_sweep(uint32_t first_bits, uint64_t n, int fused, uint32_t *first_bad) {
  uint64_t bad = 0;
  uint32_t fb = 0xFFFFFFFFu;
#pragma omp parallel for schedule(static) reduction(+ : bad) reduction(min : fb)
  for (long long t = 0; t < (long long)n; t++) {
    const uint32_t b = first_bits + (uint32_t)t;
    float x;
    memcpy(&x, &b, 4);
    const float h = logf(x), r = orc_logf_glibc(x, fused);
    if ((h != h) != (r != r) || (h == h && memcmp(&h, &r, 4))) {
      bad++;
      if (b < fb) fb = b;
    }
  }
  if (first_bad) *first_bad = fb;
  return bad;
}

/* bit patterns in the range on which the two restated forms differ (how often a host without FMA would disagree) */
uint64_t orc_logf_forms_differ(uint32_t first_bits, uint64_t n, uint32_t *out, size_t cap) {
  uint64_t bad = 0;
  for (uint64_t t = 0; t < n; t++) {
    const uint32_t b = first_bits + (uint32_t)t;
    float x;
    memcpy(&x, &b, 4);
    const float f = orc_logf_glibc(x, 1), u = orc_logf_glibc(x, 0);
    if (memcmp(&f, &u, 4) && !(f != f && u != u)) {
      if (bad < cap) out[bad] = b;
      bad++;
    }
  }
  return bad;
}

void orc_logf_array(const float *x, size_t n, float *out, int form) {
#pragma omp parallel for schedule(static)
  for (long long t = 0; t < (long long)n; t++)
    out[t] = form < 0 ? logf(x[t]) : orc_logf_glibc(x[t], form);
}

void orc_ani_from_dots(const int32_t *dot, const int32_t *nr, const int32_t *nq, size_t n, unsigned ksize, float *out) {
#pragma omp parallel for schedule(static)
  for (long long t = 0; t < (long long)n; t++) out[t] = orc_ani_from_dot(dot[t], nr[t], nq[t], ksize);
}

void orc_set_threads(int n) {
#ifdef _OPENMP
  if (n > 0) omp_set_num_threads(n);
#else
  (void)n;
#endif
}

void orc_ani_matrix(const int16_t *ref_hv, const int32_t *ref_norm2, size_t R,
                    const int16_t *qry_hv, const int32_t *qry_norm2, size_t Q,
                    size_t hv_d, unsigned ksize, float *ani_out) {
#pragma omp parallel for schedule(static)
  for (long i = 0; i < (long)R; i++)
    for (size_t j = 0; j < Q; j++) {
      int32_t dot = orc_hv_dot(ref_hv + (size_t)i * hv_d, qry_hv + j * hv_d, hv_d);
      ani_out[(size_t)i * Q + j] =
          orc_ani_from_dot(dot, ref_norm2[i], qry_norm2[j], ksize);
    }
}

/* ------------------------------------------------------------------------- */
/* bit-packed hypervectors (extension, BASELINE configs[4]; no reference code) */
/* ------------------------------------------------------------------------- */

void orc_binarize(const int16_t *hv, size_t n, size_t hv_d, uint32_t *bits) {
  size_t words = (hv_d + 31) / 32;
  for (size_t g = 0; g < n; g++)
    for (size_t w = 0; w < words; w++) {
      uint32_t v = 0;
      for (unsigned j = 0; j < 32 && w * 32 + j < hv_d; j++)
        if (hv[g * hv_d + w * 32 + j] >= 0) v |= 1u << j;
      bits[g * words + w] = v;
    }
}

void orc_hamming_matrix(const uint32_t *ref_bits, size_t R, const uint32_t *qry_bits, size_t Q,
                        size_t words, uint32_t *dist_out) {
#pragma omp parallel for schedule(static)
  for (long i = 0; i < (long)R; i++)
    for (size_t j = 0; j < Q; j++) {
      uint32_t d = 0;
      for (size_t w = 0; w < words; w++)
        d += (uint32_t)__builtin_popcount(ref_bits[(size_t)i * words + w] ^ qry_bits[j * words + w]);
      dist_out[(size_t)i * Q + j] = d;
    }
}

/* ------------------------------------------------------------------------- */
/* per-genome sketch                                                          */
/* ------------------------------------------------------------------------- */

int orc_sketch_genome(const uint8_t *seq, size_t n_bps, unsigned ksize,
                      uint64_t scaled, uint64_t seed, int canonical,
                      int norm_mode, size_t hv_d, int layout, int16_t *hv,
                      int32_t *norm2, uint32_t *n_hash) {
  uint64_t threshold = UINT64_MAX / scaled; /* src/sketch.rs:73 */
  size_t cap = n_bps / (scaled ? scaled : 1) * 2 + 1024;
  uint64_t *hs = (uint64_t *)malloc(cap * sizeof(uint64_t));
  if (!hs) return -1;
  size_t n = orc_kmer_hash_sample(seq, n_bps, ksize, threshold, seed, canonical,
                                  norm_mode, hs, cap);
  if (n > cap) { /* pathological input: redo with exact size */
    cap = n;
    uint64_t *h2 = (uint64_t *)realloc(hs, cap * sizeof(uint64_t));
    if (!h2) {
      free(hs);
      return -1;
    }
    hs = h2;
    n = orc_kmer_hash_sample(seq, n_bps, ksize, threshold, seed, canonical,
                             norm_mode, hs, cap);
  }
  n = orc_sort_unique_u64(hs, n);
  orc_encode_hv(hs, n, hv_d, layout, hv);
  *norm2 = orc_hv_norm2(hv, hv_d);
  *n_hash = (uint32_t)n;
  free(hs);
  return 0;
}

/* Task-parallel over genomes like the rayon loop of src/sketch.rs:35 (one genome per task,
 * per-thread scratch reused across genomes).  hv: n x hv_d.  Returns 0 on success. */
int orc_sketch_batch_mt(const uint8_t *const *seqs, const size_t *lens, size_t n,
                        unsigned ksize, uint64_t scaled, uint64_t seed, int canonical,
                        int norm_mode, size_t hv_d, int layout, int n_threads,
                        int16_t *hv, int32_t *norm2, uint32_t *n_hash) {
  size_t max_len = 0;
  for (size_t g = 0; g < n; g++) max_len = lens[g] > max_len ? lens[g] : max_len;
  const uint64_t threshold = UINT64_MAX / scaled;
  int fail = 0;
#pragma omp parallel num_threads(n_threads > 0 ? n_threads : 1)
  {
    uint8_t *fwd = (uint8_t *)malloc(max_len + 1), *rc = (uint8_t *)malloc(max_len + 1);
    size_t cap = max_len / (scaled ? scaled : 1) * 2 + 1024;
    uint64_t *hs = (uint64_t *)malloc(cap * sizeof(uint64_t));
    if (!fwd || !rc || !hs) {
#pragma omp atomic write
      fail = 1;
    } else {
#pragma omp for schedule(dynamic, 1)
      for (long g = 0; g < (long)n; g++) {
        size_t m = 0;
        if (lens[g] >= ksize)
          m = kmer_sample_core(seqs[g], lens[g], ksize, threshold, seed, canonical, norm_mode, hs, cap, fwd, rc);
        if (m > cap) { /* repeats-heavy input: grow and redo */
          uint64_t *h2 = (uint64_t *)realloc(hs, m * sizeof(uint64_t));
          if (!h2) {
#pragma omp atomic write
            fail = 1;
            continue;
          }
          hs = h2, cap = m;
          m = kmer_sample_core(seqs[g], lens[g], ksize, threshold, seed, canonical, norm_mode, hs, cap, fwd, rc);
        }
        m = orc_sort_unique_u64(hs, m);
        orc_encode_hv(hs, m, hv_d, layout, hv + (size_t)g * hv_d);
        norm2[g] = orc_hv_norm2(hv + (size_t)g * hv_d, hv_d);
        n_hash[g] = (uint32_t)m;
      }
    }
    free(fwd), free(rc), free(hs);
  }
  return fail ? -1 : 0;
}

/* fills n genomes (ids first .. first+n-1) in parallel; out: n x (L+1) bytes */
void orc_synth_genomes_mt(uint64_t first, size_t n, size_t L, unsigned cluster_size,
                          uint32_t sub_ppm_per_member, int n_threads, uint8_t *out) {
#pragma omp parallel for schedule(dynamic, 1) num_threads(n_threads > 0 ? n_threads : 1)
  for (long g = 0; g < (long)n; g++)
    orc_synth_genome(first + (uint64_t)g, L, cluster_size, sub_ppm_per_member, out + (size_t)g * (L + 1));
}

/* ------------------------------------------------------------------------- */
/* synthetic genomes                                                          */
/* ------------------------------------------------------------------------- */

static inline uint64_t splitmix64(uint64_t x) {
  uint64_t z = x + 0x9e3779b97f4a7c15ull;
  z = (z ^ (z >> 30)) * 0xbf58476d1ce4e5b9ull;
  z = (z ^ (z >> 27)) * 0x94d049bb133111ebull;
  return z ^ (z >> 31);
}

void orc_synth_genome(uint64_t g, size_t L, unsigned cluster_size,
                      uint32_t sub_ppm_per_member, uint8_t *out) {
  static const char ACGT[4] = {'A', 'C', 'G', 'T'};
  uint64_t c = g / cluster_size, m = g % cluster_size;
  uint64_t key_root = splitmix64(0x48595045ull + c);
  uint64_t key_mut = splitmix64(0x4d555441ull + g);
  /* substitution probability = m * ppm / 1e6, as a 32-bit threshold */
  uint64_t thr = (m * (uint64_t)sub_ppm_per_member * 4294967296ull) / 1000000ull;
  out[0] = 'N';
  for (size_t p = 0; p < L; p++) {
    uint64_t w = splitmix64(key_root + (p >> 5));
    unsigned code = (unsigned)(w >> (2 * (p & 31))) & 3;
    if (thr) {
      uint64_t u = splitmix64(key_mut + p);
      if ((u >> 32) < thr) code = (code + 1 + (unsigned)((u & 0xffff) % 3)) & 3;
    }
    out[1 + p] = (uint8_t)ACGT[code];
  }
}
