// hg_kmer_kernels.hip -- FracMinHash k-mer hash + threshold sample on gfx950.
//
// Computes what extract_kmer_hash (src/sketch.rs:71-98) / cuda_kmer_t1ha2
// (src/cuda_kernel.cu:250-321) compute: for every k-window of ACGT bases the canonical
// strand's ASCII bytes are hashed with t1ha2_atonce(seed) and hashes below the threshold
// are kept.  The design is MI355X-first, not the reference's (1 thread x 512 k-mers with
// two 544-byte scratch arrays):
//
//  * one lane owns a 32-base register window and the M = (33-k)&~3 k-mers that start in
//    its first M bases (k <= 21; k = 22..32: a 64-base window and 32 starts); a 256-lane
//    workgroup walks 8 consecutive tiles, so one 5 Mbp genome is ~200 workgroups and a
//    1000-genome batch fills the 256 CUs many times;
//  * bases are classified 4 at a time (SWAR on dwords): 2-bit codes, upper-cased ASCII
//    and complement ASCII come from v_perm_b32 lookups, validity from one XOR;
//  * the canonical strand is chosen by ONE 64-bit compare of 2-bit packed k-mers
//    (A<C<G<T holds both in ASCII and in the 2-bit code, so this equals the reference's
//    byte-wise compare, src/cuda_kernel.cu:306-311);
//  * the hash input words are cut out of the register window with v_alignbyte_b32 /
//    v_perm_b32 (constant selectors), only the chosen strand is hashed, nothing touches
//    scratch;
//  * t1ha2 is specialised at compile time for k (17..24 => 3 mixups + final = 4 128-bit
//    products + 2 64-bit products = 22 v_mad_u64_u32 / v_mul_lo_u32);
//  * survivors (1/scaled of the k-mers) are staged in a small LDS list and the workgroup
//    reserves its range of the genome's hit slice with ONE global atomic at the end of its
//    work item (a returning global atomic per hit parked the wave for a memory round trip);
//    lossless: no 8-slot cap like src/cuda_kernel.cu:316, hash value 0 is kept.
//
// The kernel is integer-VALU bound (~100 lane-instructions per k-mer), not HBM bound.
#include <cstdlib>
#include <utility>

#include "hg_internal.h"

namespace {

// t1ha2 primes (src/cuda_kernel.cu:71-77)
constexpr uint64_t P0 = 0xEC99BF0D8372CAABull;
constexpr uint64_t P1 = 0x82434FE90EDCEF39ull;
constexpr uint64_t P2 = 0xD4F06DB99D67BE4Bull;
constexpr uint64_t P3 = 0xBD9CACC22C6E9571ull;
constexpr uint64_t P4 = 0x9C06FAF4D023E3ABull;
constexpr uint64_t P5 = 0xC060724A8424F345ull;
constexpr uint64_t P6 = 0xCB5AF53AE3AAAC31ull;

__device__ __forceinline__ uint64_t mk64(uint32_t lo, uint32_t hi) {
  return (uint64_t)lo | ((uint64_t)hi << 32);
}
#ifndef HG_ROT_ALIGNBIT
#define HG_ROT_ALIGNBIT 1  /* rot64 by a constant as two v_alignbit_b32 (0: the compiler's 64-bit shift + shift + or) */
#endif
__device__ __forceinline__ uint64_t rot64(uint64_t v, unsigned s) {  // rotate right
#if HG_ROT_ALIGNBIT
  if (__builtin_constant_p(s) && s > 0 && s < 64 && s != 32) {
    const uint32_t lo = (uint32_t)v, hi = (uint32_t)(v >> 32);
    return s < 32 ? mk64(__builtin_amdgcn_alignbit(hi, lo, s), __builtin_amdgcn_alignbit(lo, hi, s))
                  : mk64(__builtin_amdgcn_alignbit(lo, hi, s - 32), __builtin_amdgcn_alignbit(hi, lo, s - 32));
  }
#endif
  return (v >> s) | (v << (64 - s));
}
// lo64(x * P) returned, hi64(x * P) + addend stored in hi_plus.
// Schoolbook product on 32-bit halves with four v_mad_u64_u32.  Written by hand because the
// compiler's expansion of a 128-bit multiply spends 5 v_mov + one 64-bit add on zero-extending
// partial words; here the third product takes the second as its 64-bit addend and hands its carry
// out in an SGPR pair, and that carry plus the caller's addend are folded into the addend of the
// last product with three 32-bit ops.
template <uint64_t P, bool ZERO_ADDEND = false, bool UNI_ADDEND = false>
__device__ __forceinline__ uint64_t mul128_lo_hiadd(uint64_t x, uint64_t addend, uint64_t &hi_plus) {
  constexpr uint32_t p0 = (uint32_t)P, p1 = (uint32_t)(P >> 32);
  const uint32_t x0 = (uint32_t)x, x1 = (uint32_t)(x >> 32);
  const uint64_t A = (uint64_t)x0 * p0;
  const uint64_t T = (uint64_t)x1 * p0 + (A >> 32);  // cannot overflow
  uint64_t W, cm;                                     // W = x0*p1 + T (mod 2^64), cm = carry-out lanes
  asm("v_mad_u64_u32 %0, %1, %2, %3, %4" : "=v"(W), "=s"(cm) : "v"(x0), "s"(p1), "v"(T));
  if (ZERO_ADDEND) {  // S = {hi32(W), carry}
    uint32_t shi;
    asm("v_cndmask_b32_e64 %0, 0, 1, %1" : "=v"(shi) : "s"(cm));
    hi_plus = (uint64_t)x1 * p1 + mk64((uint32_t)(W >> 32), shi);
  } else {
    // hi64 + addend = x1*p1 + (addend + hi32(W)) + (carry << 32): the 64-bit sum addend + hi32(W) is ONE
    // v_mad_u64_u32 (hi32(W) * 1 + addend; it lands in an aligned register pair, which three 32-bit carry ops
    // do not -- the compiler then paid v_movs to pair them), and the carry goes into the high half of the last
    // product with one v_addc that takes it straight from the SGPR pair.  All of this is mod 2^64, like b += hi.
    // (UNI_ADDEND: the addend is wave-uniform -- the seed in the first mixup -- and is read from its SGPR pair)
    uint64_t t, junk;
    if (UNI_ADDEND)
      asm("v_mad_u64_u32 %0, %1, %2, 1, %3" : "=v"(t), "=s"(junk) : "v"((uint32_t)(W >> 32)), "s"(addend));
    else
      asm("v_mad_u64_u32 %0, %1, %2, 1, %3" : "=v"(t), "=s"(junk) : "v"((uint32_t)(W >> 32)), "v"(addend));
    const uint64_t hp = (uint64_t)x1 * p1 + t;
    uint32_t hph = (uint32_t)(hp >> 32);
    asm("v_addc_co_u32_e64 %0, %1, %0, 0, %1" : "+v"(hph), "+s"(cm));
    hi_plus = mk64((uint32_t)hp, hph);
  }
  return mk64((uint32_t)A, (uint32_t)W);
}

// src/cuda_kernel.cu:136-141 with the prime as a template argument
template <uint64_t P, bool HAND = true, bool UNI_B = false>
__device__ __forceinline__ void mixup64(uint64_t &a, uint64_t &b, uint64_t v) {
  if (HAND) {
    uint64_t nb;
    a ^= mul128_lo_hiadd<P, false, UNI_B>(b + v, b, nb);
    b = nb;
  } else {
    unsigned __int128 m = (unsigned __int128)(b + v) * P;
    a ^= (uint64_t)m;
    b += (uint64_t)(m >> 64);
  }
}
// lo64(x * P).  The compiler's form is one v_mad_u64_u32 + two v_mul_lo_u32 + v_add3 (four slow-class
// instructions); here the two cross products are chained through one 64-bit accumulator -- three v_mad_u64_u32 and a
// plain add (HG_LO64MUL_HAND=0: the compiler's form, A/B).
#ifndef HG_LO64MUL_HAND
#define HG_LO64MUL_HAND 1
#endif
template <uint64_t P>
__device__ __forceinline__ uint64_t lo64mul(uint64_t x) {
#if HG_LO64MUL_HAND
  constexpr uint32_t p0 = (uint32_t)P, p1 = (uint32_t)(P >> 32);
  const uint32_t x0 = (uint32_t)x, x1 = (uint32_t)(x >> 32);
  const uint64_t t = (uint64_t)x0 * p0;
  uint64_t w1, w, junk;
  asm("v_mad_u64_u32 %0, %1, %2, %3, 0" : "=v"(w1), "=s"(junk) : "v"(x0), "s"(p1));
  asm("v_mad_u64_u32 %0, %1, %2, %3, %4" : "=v"(w), "=s"(junk) : "v"(x1), "s"(p0), "v"(w1));
  return mk64((uint32_t)t, (uint32_t)(t >> 32) + (uint32_t)w);
#else
  return x * P;
#endif
}
// src/cuda_kernel.cu:143-153
template <bool HAND = true>
__device__ __forceinline__ uint64_t final64(uint64_t a, uint64_t b) {
  uint64_t x = HAND ? lo64mul<P0>(a + rot64(b, 41)) : (a + rot64(b, 41)) * P0;
  uint64_t y = HAND ? lo64mul<P6>(rot64(a, 23) + b) : (rot64(a, 23) + b) * P6;
  if (HAND) {
    uint64_t hi;
    const uint64_t lo = mul128_lo_hiadd<P5, true>(x ^ y, 0, hi);
    return lo ^ hi;
  }
  unsigned __int128 m = (unsigned __int128)(x ^ y) * P5;
  return (uint64_t)m ^ (uint64_t)(m >> 64);
}

// t1ha2_atonce for a compile-time length K <= 32 whose bytes are given as little-endian
// dwords d[0..ceil(K/4)) with the unused bytes of the last dword zero
// (tail switch of src/cuda_kernel.cu:205-245).
template <int K, bool HAND = true>
__device__ __forceinline__ uint64_t t1ha2_fixed(const uint32_t *d, uint64_t seed) {
  constexpr int ND = (K + 3) / 4;
  auto word = [&](int i) -> uint64_t {  // i-th 8-byte word, zero padded
    uint32_t lo = (2 * i < ND) ? d[2 * i] : 0u;
    uint32_t hi = (2 * i + 1 < ND) ? d[2 * i + 1] : 0u;
    return mk64(lo, hi);
  };
  uint64_t a = seed, b = (uint64_t)K;
  int i = 0;
  if (K > 24) mixup64<P4, HAND>(a, b, word(i++));
  if (K > 16) mixup64<P3, HAND>(b, a, word(i++));
  if (K > 8) mixup64<P2, HAND>(a, b, word(i++));
  if (K > 0) mixup64<P1, HAND>(b, a, word(i++));
  return final64<HAND>(a, b);
}

// the same on 8-byte words w[0..ceil(K/8)) (unused bytes of the last word zero)
template <int K>
__device__ __forceinline__ uint64_t t1ha2_fixed_w(const uint64_t *w, uint64_t seed) {
  // the first mixup's `b` operand is wave-uniform (the seed or the length)
  uint64_t a = seed, b = (uint64_t)K;
  int i = 0;
  if (K > 24) mixup64<P4, true, true>(a, b, w[i++]);
  if (K > 16) mixup64<P3, true, (K <= 24)>(b, a, w[i++]);
  if (K > 8) mixup64<P2, true, (K <= 16)>(a, b, w[i++]);
  if (K > 0) mixup64<P1, true, (K <= 8)>(b, a, w[i++]);
  return final64(a, b);
}

template <int... Js, class F>
__device__ __forceinline__ void static_for(std::integer_sequence<int, Js...>, F &&f) {
  (f(std::integral_constant<int, Js>{}), ...);
}

// ---- geometry ---------------------------------------------------------------------------
constexpr int WG = 256;        // lanes per workgroup
// tiles of a work item: 8, or 9 where a lane owns 12 starts per tile (k = 18..21) -- the grouped kernel walks the
// same item as 3 tiles of 3 x 12 starts per lane
constexpr int tiles_per_item(int k) { return (((33 - k) & ~3) == 12) ? 9 : 8; }
template <int K>
struct Geo {
  static constexpr int M = (33 - K) & ~3;  // k-mer starts per lane, multiple of 4 (dword stride)
  static constexpr int TILES_PER_ITEM = tiles_per_item(K);
  static constexpr int ND = (K + 3) / 4;   // dwords of one k-mer
  static constexpr int NB = K - 4 * (ND - 1);  // bytes used in the last dword (1..4)
  static constexpr int TILE = WG * M;
  static constexpr int ITEM = TILE * TILES_PER_ITEM;
};
constexpr int GEN_STARTS = 32;                       // long-k kernel: starts per lane
constexpr int GEN_ITEM = WG * GEN_STARTS;            // and per work item
// k >= FAST64_FROM runs the 64-base-window kernel: with 32-base windows a lane owns only (33 - k) & ~3 k-mer
// starts (8 at k = 22..25, 4 at k = 26..29).  Measured A/B on one box: k = 22..25 4-6 % faster, k = 26..29
// 16-17 % faster with the wide window; k <= 21 the two kernels tie.
constexpr uint32_t FAST64_FROM = 22;
constexpr int TILES_PER_ITEM64 = 4;  // work item of k = 22..32: 4 tiles x 256 lanes x 32 starts
constexpr bool fast64_k(uint32_t k) { return k >= FAST64_FROM && k <= 32; }
// canonical fast kernels: 4 = chosen strand fetched from an LDS image of the window (| 8: dword-aligned reads +
// run-time v_alignbyte; without it byte-offset ds_reads; | 16: a second copy of the k-mer loop without the validity
// test for waves that saw only bases), 0 = register extraction of both strands + mux
#ifndef HG_KMER_DEFAULT_VAR
#define HG_KMER_DEFAULT_VAR 28
#endif
#ifndef HG_U2T_HOIST
#define HG_U2T_HOIST 1  /* the u/U -> T rewrite behind one branch per window (0: one branch per dword) */
#endif
#ifndef HG_KMER_SHARED
#define HG_KMER_SHARED 1  /* k = 1..21: kmer_sample_shared (workgroup-wide phase images in LDS) instead of kmer_sample_grouped / _fast (A/B: -DHG_KMER_SHARED=0) */
#endif
#ifndef HG_KMER_GROUPED
#define HG_KMER_GROUPED 1  /* canonical k = 18..25: kmer_sample_grouped instead of kmer_sample_fast / fast64 (A/B: -DHG_KMER_GROUPED=0) */
#endif
constexpr bool fast_k(uint32_t k) { return k >= 1 && k < FAST64_FROM; }

__device__ __forceinline__ void append_hit(uint64_t h, const hg_genome_meta &gm, uint32_t g,
                                           uint64_t *__restrict__ hits, uint32_t *__restrict__ cnt) {
  uint32_t idx = atomicAdd(&cnt[g], 1u);  // hipcc folds this into one atomic per wave
  if (idx < gm.hit_cap) hits[gm.hit_off + idx] = h;
}

// Workgroup-level staging of the sampled hashes (fast kernels).  A hit is 1 k-mer in `scaled`, but every
// one used to cost its wave a returning global atomic -- a full memory round trip in the middle of ~1 400
// VALU instructions, about once per two tiles per wave.  Hits go to an LDS list instead (LDS atomic) and the
// workgroup reserves its range of the genome's hit buffer once, at the end of its work item.  A list that
// overflows (low-complexity sequence: every position of a repeat samples the same hash) spills straight to
// the global path; the raw counter semantics (it keeps counting past the capacity) are unchanged.
constexpr uint32_t HIT_STAGE = 256;
struct HitStage {
  uint64_t h[HIT_STAGE];
  uint32_t n, base;
};
__device__ __forceinline__ void stage_hit(HitStage &st, uint64_t h, const hg_genome_meta &gm, uint32_t g,
                                          uint64_t *__restrict__ hits, uint32_t *__restrict__ cnt) {
  const uint32_t idx = atomicAdd(&st.n, 1u);
  if (idx < HIT_STAGE) st.h[idx] = h;
  else append_hit(h, gm, g, hits, cnt);
}
__device__ __forceinline__ void flush_hits(HitStage &st, const hg_genome_meta &gm, uint32_t g,
                                           uint64_t *__restrict__ hits, uint32_t *__restrict__ cnt) {
  __syncthreads();
  const uint32_t n = st.n < HIT_STAGE ? st.n : HIT_STAGE;
  if (threadIdx.x == 0 && n) st.base = atomicAdd(&cnt[g], n);
  __syncthreads();
  for (uint32_t i = threadIdx.x; i < n; i += blockDim.x) {
    const uint32_t idx = st.base + i;
    if (idx < gm.hit_cap) hits[gm.hit_off + idx] = st.h[i];
  }
}

// =========================================================================================
// fast kernel: compile-time k in [1, 21] (instantiable up to 29)
// =========================================================================================
template <int K, bool CANON, int VAR = 0>
__global__ __launch_bounds__(WG) void kmer_sample_fast(
    const uint8_t *__restrict__ seq, const hg_genome_meta *__restrict__ meta,
    const uint32_t *__restrict__ item_genome, uint64_t threshold, uint64_t seed, uint32_t u2t,
    uint64_t *__restrict__ hits, uint32_t *__restrict__ cnt) {
  using G = Geo<K>;
  constexpr int M = G::M, ND = G::ND, NB = G::NB;
  constexpr uint64_t MASK2K = (K == 32) ? ~0ull : ((1ull << (2 * K)) - 1);
  constexpr uint32_t MASKK = (1u << K) - 1;

  const uint32_t item = blockIdx.x;
  const uint32_t g = item_genome[item];
  const hg_genome_meta gm = meta[g];
  const uint64_t n_bps = gm.n_bps;
  if (n_bps < (uint64_t)K) return;
  const uint64_t n_starts = n_bps - K + 1;
  const uint8_t *__restrict__ gseq = seq + gm.seq_off;
  const uint64_t item_start = (uint64_t)(item - gm.item_first) * G::ITEM;
  __shared__ HitStage stage;
  // VAR & 4 (canonical strand only): the lane's window goes to LDS twice -- forward ASCII F[0..32) and its reverse
  // complement R[i] = comp(F[31 - i]) -- so that BOTH strands of k-mer j are ascending byte ranges of one 64-byte
  // image (forward: [j, j+K), reverse: [64-K-j, 64-j)) and the chosen strand's hash words are fetched with
  // (unaligned) ds_read_b64 from a run-time offset.  That replaces, per k-mer, the v_alignbyte / v_perm
  // extraction of both strands and the six-dword v_bitop3 mux (~12 VALU instructions) by one v_cndmask + one
  // add and three LDS reads, which issue beside the VALU stream.  Lanes only read what they wrote themselves:
  // no barrier.  Lane pitch: 68 bytes = 17 dwords for the dword-aligned variant (VAR & 8) -- its ds_read2_b32 /
  // ds_write2_b32 are banked over 32 banks per half wave, and 17 is odd (the first version used 72 bytes: 18 l mod 32
  // repeats after 16 lanes, every access was a 2-way conflict and SQ_LDS_BANK_CONFLICT was 68 % of the LDS-active
  // cycles, with the LDS pipe busy 60 % of the kernel); 72 bytes for the byte-offset variant (b64 / b128 accesses).
  // The overshoot of the last word stays inside the lane's own pad.
  constexpr bool LDSWIN = CANON && (VAR & 4) != 0;
  constexpr int WIN_PITCH = (VAR & 8) ? 68 : 72;
  __shared__ __attribute__((aligned(16))) uint8_t s_win[LDSWIN ? WG * WIN_PITCH + 16 : 16];
  uint8_t *const mywin = s_win + threadIdx.x * WIN_PITCH;
  if (threadIdx.x == 0) stage.n = 0;
  __syncthreads();

  // the lane's 32-base window: 8 dwords, 4-byte aligned, M-byte lane stride.  Lanes past the
  // genome end produce nothing (inv = all ones below): they are pointed at the genome start so
  // that they never read beyond the 32-byte slack.
  uint32_t xn[8];
  auto load_window = [&](uint64_t tile_start_) {
    const uint64_t p = tile_start_ + (uint64_t)threadIdx.x * M;
    const uint32_t *src = reinterpret_cast<const uint32_t *>(gseq + (p < n_bps ? p : 0));
#pragma unroll
    for (int t = 0; t < 8; ++t) xn[t] = src[t];
  };

  // VAR & 32: the next tile's window is requested while the current one is hashed (the LDS-image variants leave
  // the registers for it)
  constexpr bool PREFETCH = (VAR & 32) != 0;
  if (PREFETCH && item_start < n_starts) load_window(item_start);
#pragma unroll 1
  for (int tile = 0; tile < G::TILES_PER_ITEM; ++tile) {
    const uint64_t tile_start = item_start + (uint64_t)tile * G::TILE;
    if (tile_start >= n_starts) break;  // uniform
    const uint64_t p0 = tile_start + (uint64_t)threadIdx.x * M;

    uint32_t x[8];
    if (!PREFETCH) load_window(tile_start);
#pragma unroll
    for (int t = 0; t < 8; ++t) x[t] = xn[t];
    if (PREFETCH && tile + 1 < G::TILES_PER_ITEM && tile_start + G::TILE < n_starts) load_window(tile_start + G::TILE);

    // ---- classify 4 bases per dword -----------------------------------------------------
    uint32_t FA[8], CA[8];      // upper-case ASCII, complement ASCII (same byte order)
    uint32_t dacc = 0;          // != 0  <=> some byte of the window is not ACGTacgt
    uint32_t Glo = 0, Ghi = 0;  // 2-bit codes, base b at bits [2b, 2b+1]
#if HG_U2T_HOIST
    if (u2t) {  // needletail normalize: u/U -> T  ('U' ^ 'T' == 1); ONE uniform branch per window
#pragma unroll
      for (int t = 0; t < 8; ++t) {
        const uint32_t e = (x[t] & 0xDFDFDFDFu) ^ 0x55555555u;
        const uint32_t nz = ((e & 0x7F7F7F7Fu) + 0x7F7F7F7Fu) | e;  // bit 7 set <=> byte != 'U'
        x[t] ^= (~nz & 0x80808080u) >> 7;
      }
    }
#endif
#pragma unroll
    for (int t = 0; t < 8; ++t) {
#if !HG_U2T_HOIST
      if (u2t) {
        const uint32_t e = (x[t] & 0xDFDFDFDFu) ^ 0x55555555u;
        const uint32_t nz = ((e & 0x7F7F7F7Fu) + 0x7F7F7F7Fu) | e;
        x[t] ^= (~nz & 0x80808080u) >> 7;
      }
#endif
      const uint32_t xv = x[t];
      uint32_t u = xv & 0xDFDFDFDFu;
      uint32_t tt = xv ^ (xv >> 1);
      uint32_t cd = (tt >> 1) & 0x03030303u;  // A,C,G,T -> 0,1,2,3 per byte
      FA[t] = __builtin_amdgcn_perm(0u, 0x54474341u, cd);  // "ACGT"[code]
      CA[t] = __builtin_amdgcn_perm(0u, 0x41434754u, cd);  // "TGCA"[code]
      dacc |= u ^ FA[t];
      // c0 | c1<<2 | c2<<4 | c3<<6 as ONE byte dot product with (1, 4, 16, 64)
      const uint32_t p = __builtin_amdgcn_udot4(cd, 0x40100401u, 0u, false);
      if (t < 4) Glo |= p << (8 * t);
      else Ghi |= p << (8 * (t - 4));
    }
    // MSB-first copy of the codes (base 0 in the top two bits) for the forward k-mer value
    auto pairrev = [](uint32_t v) {
      uint32_t br = __builtin_bitreverse32(v);
      return ((br >> 1) & 0x55555555u) | ((br & 0x55555555u) << 1);
    };
    const uint64_t Gl = mk64(Glo, Ghi);
    const uint64_t Gm = mk64(pairrev(Ghi), pairrev(Glo));
    const uint64_t Gc = ~Gl;  // complement codes; read LSB-first this IS the reverse strand

    // ---- validity: rare path, taken only by waves that see a non-base or the genome end ----
    const int64_t rem64 = (int64_t)n_bps - (int64_t)p0;
    const uint32_t rem = rem64 >= 32 ? 32u : (rem64 <= 0 ? 0u : (uint32_t)rem64);
    uint32_t inv = 0;  // bit b set <=> base b of the window cannot be part of a k-mer
    const bool wave_dirty = __any((dacc != 0) | (rem < 32));  // wave-uniform
    if (wave_dirty) {
#pragma unroll
      for (int t = 0; t < 8; ++t) {
        const uint32_t xv = x[t];  // (u/U already turned into T above)
        uint32_t d = (xv & 0xDFDFDFDFu) ^ FA[t];
        uint32_t z = (((d & 0x7F7F7F7Fu) + 0x7F7F7F7Fu) | d) & 0x80808080u;
        uint32_t nib = (((z >> 7) * 0x01020408u) >> 24) & 0xFu;
        inv |= nib << (4 * t);
      }
      if (rem < 32) inv |= (rem == 0) ? ~0u : (~0u << rem);
    }

    if (LDSWIN) {
      // R dword t = bytes comp(F[31-4t]), comp(F[30-4t]), ... = CA[7 - t] byte-reversed
      if constexpr ((VAR & 8) != 0) {
        uint32_t *w1 = reinterpret_cast<uint32_t *>(mywin);  // 4-byte aligned pitch: dword stores (paired by the compiler)
#pragma unroll
        for (int t = 0; t < 8; ++t) {
          w1[t] = FA[t];
          w1[8 + t] = __builtin_amdgcn_perm(0u, CA[7 - t], 0x00010203u);
        }
      } else {
        uint2 *w2 = reinterpret_cast<uint2 *>(mywin);
#pragma unroll
        for (int t = 0; t < 4; ++t) {
          w2[t] = make_uint2(FA[2 * t], FA[2 * t + 1]);
          w2[4 + t] = make_uint2(__builtin_amdgcn_perm(0u, CA[7 - 2 * t], 0x00010203u),
                                 __builtin_amdgcn_perm(0u, CA[6 - 2 * t], 0x00010203u));
        }
      }
    }

    // strand choice + hash words of k-mer jj (LDS image variant)
    constexpr int NW = (K + 7) / 8, TB = K - 8 * (NW - 1);  // 8-byte words of a k-mer, bytes used in the last one
    uint64_t wq[2][NW];
    auto fetch_words = [&](auto jjc, uint64_t *w) {
      constexpr int jj = decltype(jjc)::value;
      uint64_t fv, rv;
      if constexpr ((K & 1) != 0) {
        // odd K: a k-mer never equals its reverse complement (the middle base would have to pair with itself), so the
        // compare is decided inside the 2K bits and the two values only have to be TOP-aligned -- the bits below them
        // (codes of neighbouring bases) never matter and need no masking
        fv = Gm << (2 * jj);
        rv = Gc << (2 * (32 - K - jj));
      } else {
        fv = (Gm >> (2 * (32 - K - jj))) & MASK2K;
        rv = (Gc >> (2 * jj)) & MASK2K;
      }
      uint64_t lt;
      uint32_t off;
      asm("v_cmp_lt_u64_e64 %0, %1, %2" : "=s"(lt) : "v"(rv), "v"(fv));
      asm("v_cndmask_b32_e64 %0, %1, %2, %3" : "=v"(off) : "n"(jj), "n"(64 - K - jj), "s"(lt));
      if constexpr ((VAR & 8) != 0) {
        // dword-aligned reads + run-time v_alignbyte (it takes the byte shift from off[1:0]): a ds_read at an
        // address that is not a multiple of 4 is executed lane by lane on gfx950 (measured: ~64 cycles per wave
        // instruction, the kernel took 16.8 ms instead of 10.3 with byte-offset ds_read_b64/b128)
        constexpr int NDR = (K + 3 + 3) / 4;  // aligned dwords that cover any K bytes starting at shift 0..3
        const uint32_t *src = reinterpret_cast<const uint32_t *>(__builtin_assume_aligned(mywin + (off & ~3u), 4));
        uint32_t t[NDR + 1], d[2 * NW];
#pragma unroll
        for (int m = 0; m < NDR; ++m) t[m] = src[m];
        t[NDR] = 0;
#pragma unroll
        for (int m = 0; m < 2 * NW; ++m) {
          if (m < ND) d[m] = __builtin_amdgcn_alignbyte(t[m + 1 < NDR ? m + 1 : NDR], t[m], off);
          else d[m] = 0;
        }
        if constexpr (NB < 4) d[ND - 1] &= (1u << (8 * (NB & 3))) - 1;
#pragma unroll
        for (int m = 0; m < NW; ++m) w[m] = mk64(d[2 * m], d[2 * m + 1]);
        return;
      }
      const uint8_t *src = mywin + off;
#pragma unroll
      for (int m = 0; m < NW - 1; ++m) __builtin_memcpy(&w[m], src + 8 * m, 8);
      if constexpr (TB <= 4) {  // last word: TB bytes
        uint32_t lo;
        __builtin_memcpy(&lo, src + 8 * (NW - 1), 4);
        if constexpr (TB < 4) lo &= (1u << (8 * (TB & 3))) - 1;
        w[NW - 1] = lo;
      } else {
        __builtin_memcpy(&w[NW - 1], src + 8 * (NW - 1), 8);
        if constexpr (TB < 8) w[NW - 1] &= (1ull << (8 * (TB & 7))) - 1;
      }
    };

    // ---- the lane's M k-mers ------------------------------------------------------------------
    // VAR & 16: two copies of the loop -- waves that saw only bases (the normal case: the validity mask is zero for
    // every lane) run one without the per-k-mer validity test
    auto run_kmers = [&](auto checkc) {
    constexpr bool CHECK = decltype(checkc)::value;
    static_for(std::make_integer_sequence<int, M>{}, [&](auto jc) {
      constexpr int j = decltype(jc)::value;
      constexpr int q = j >> 2, r = j & 3;
      const bool valid = !CHECK || ((inv >> j) & MASKK) == 0;

      if constexpr (LDSWIN) {
        // two-deep software pipeline: the words of k-mer j+1 are requested before k-mer j is hashed
        if constexpr (j == 0) fetch_words(jc, wq[0]);
        if constexpr (j + 1 < M) fetch_words(std::integral_constant<int, j + 1>{}, wq[(j + 1) & 1]);
        const uint64_t h = t1ha2_fixed_w<K>(wq[j & 1], seed);
        if (valid && h < threshold) stage_hit(stage, h, gm, g, hits, cnt);
        return;
      }

      // strand choice as an all-ones / all-zeros VGPR mask.  The obvious `use_rc ? rcw : f` becomes
      // v_cmp + 6 x v_cndmask_b32 ... vcc, and that VOP2/VCC form measures ~14-20 cycles per
      // instruction on gfx950 (tools/gpu_microbench.hip); the compare result is therefore taken into
      // an SGPR pair once, turned into a mask, and the six dwords are muxed with v_bitop3_b32.
      uint32_t rc_mask = 0;
      bool use_rc = false;
      if (CANON) {
        const uint64_t fv = (Gm >> (2 * (32 - K - j))) & MASK2K;
        const uint64_t rv = (Gc >> (2 * j)) & MASK2K;
        if (VAR & 2) {
          use_rc = rv < fv;
        } else {
          uint64_t lt;
          asm("v_cmp_lt_u64_e64 %0, %1, %2" : "=s"(lt) : "v"(rv), "v"(fv));
          asm("v_cndmask_b32_e64 %0, 0, -1, %1" : "=v"(rc_mask) : "s"(lt));
        }
      }

      uint32_t d[ND];
#pragma unroll
      for (int m = 0; m < ND; ++m) {
        // forward strand: bytes j+4m .. j+4m+3 of FA
        uint32_t f;
        if (m < ND - 1) {
          f = (r == 0) ? FA[q + m] : __builtin_amdgcn_alignbyte(FA[q + m + 1], FA[q + m], r);
        } else if (r + NB <= 4) {
          f = (NB == 4) ? FA[q + m] : ((FA[q + m] >> (8 * r)) & ((1u << (8 * (NB & 3))) - 1));
        } else {
          f = __builtin_amdgcn_alignbyte(FA[q + m + 1], FA[q + m], r);
          if (NB < 4) f &= (1u << (8 * (NB & 3))) - 1;
        }
        uint32_t v = f;
        if (CANON) {
          // reverse strand: byte i of this dword is comp(base[e - i]), e = j+K-1-4m
          const int e = j + K - 1 - 4 * m;
          const int Q = e >> 2, s = e & 3;
          uint32_t sel = 0;
#pragma unroll
          for (int i = 0; i < 4; ++i) {
            uint32_t sb = (4 * m + i < K) ? (uint32_t)(s + 4 - i) : 0x0cu;  // 0x0c -> 0x00
            sel |= sb << (8 * i);
          }
          uint32_t rcw = __builtin_amdgcn_perm(CA[Q], (Q >= 1) ? CA[Q - 1] : 0u, sel);
          v = (VAR & 2) ? (use_rc ? rcw : f)
                        : __builtin_amdgcn_bitop3_b32(rc_mask, rcw, f, 0xCA);  // rc_mask ? rcw : f
        }
        d[m] = v;
      }
      const uint64_t h = t1ha2_fixed<K, !(VAR & 1)>(d, seed);
      if (valid && h < threshold) stage_hit(stage, h, gm, g, hits, cnt);
    });
    };
    if ((VAR & 16) != 0 && !wave_dirty) run_kmers(std::false_type{});
    else run_kmers(std::true_type{});
  }
  flush_hits(stage, gm, g, hits, cnt);
}

// =========================================================================================
// grouped kernel: canonical k = 18..21 (12 k-mer starts per 32-base window)
// =========================================================================================
// kmer_sample_fast spends 12.5 of its 75 VALU instructions per k-mer on what happens once per tile and lane -- loading
// and classifying the 32-base window, packing its 2-bit codes, writing the LDS image -- for only 12 k-mers, whose
// windows overlap their neighbours' by 20 bases.  Here a lane owns a 56-base window and three GROUPS of 12 k-mers:
// the window is loaded, classified and written to LDS (forward image + reverse complement, 112 bytes) once, and each
// group runs the same 12-k-mer body on its own 32-base slice -- the code words of the slice come out of the window's
// 112-bit code streams with four v_alignbit each, its LDS offsets are the image offsets -/+ 12 g bytes (a multiple
// of 4: the run-time byte shift of the hash words does not change), taken through one selected base pointer.
// The group loop is a real loop (three unrolled copies of 2 x 12 hash bodies would not fit the instruction cache).
template <int K>
__global__ __launch_bounds__(WG) void kmer_sample_grouped(
    const uint8_t *__restrict__ seq, const hg_genome_meta *__restrict__ meta,
    const uint32_t *__restrict__ item_genome, uint64_t threshold, uint64_t seed, uint32_t u2t,
    uint64_t *__restrict__ hits, uint32_t *__restrict__ cnt) {
  using G = Geo<K>;
  static_assert(G::M == 12 || G::M == 8, "12 starts per slice (k = 18..21) or 8 (k = 22..25)");
  // 3 slices of 12 or 4 slices of 8 starts: 36 / 32 starts per lane and window; 56 bases = 14 dwords either way
  constexpr int M = G::M, GROUPS = 24 / M + 1, MB = M * GROUPS, BW = (GROUPS - 1) * M + 32, NDW = BW / 4;
  static_assert(BW == 56, "window of 56 bases");
  constexpr int ND = G::ND, NB = G::NB, NW = (K + 7) / 8;
  constexpr int PITCH = 2 * BW + 4;  // 116 bytes = 29 dwords: odd, the lanes' dword accesses spread over the banks
  // the work item is the one the host plans for this k (hg_kmer_item_starts): k >= 22 shares kmer_sample_fast64's
  constexpr int ITEM = fast64_k(K) ? WG * 32 * TILES_PER_ITEM64 : G::ITEM;
  constexpr int TILE = WG * MB, TILES = ITEM / TILE;
  static_assert(TILES * TILE == ITEM, "the work item is a whole number of grouped tiles");
  constexpr uint32_t MASKK = (1u << K) - 1;

  const uint32_t item = blockIdx.x;
  const uint32_t g = item_genome[item];
  const hg_genome_meta gm = meta[g];
  const uint64_t n_bps = gm.n_bps;
  if (n_bps < (uint64_t)K) return;
  const uint64_t n_starts = n_bps - K + 1;
  const uint8_t *__restrict__ gseq = seq + gm.seq_off;
  const uint64_t item_start = (uint64_t)(item - gm.item_first) * ITEM;
  __shared__ HitStage stage;
  __shared__ __attribute__((aligned(16))) uint8_t s_win[WG * PITCH + 16];
  uint8_t *const mywin = s_win + threadIdx.x * PITCH;
  if (threadIdx.x == 0) stage.n = 0;
  __syncthreads();

#pragma unroll 1
  for (int tile = 0; tile < TILES; ++tile) {
    const uint64_t tile_start = item_start + (uint64_t)tile * TILE;
    if (tile_start >= n_starts) break;  // uniform
    const uint64_t p0 = tile_start + (uint64_t)threadIdx.x * MB;
    uint32_t x[NDW];
    {
      // 56 bytes; a window that would run more than 32 bytes past the genome end (the slack every caller provides) is
      // read from the genome start instead and, if it holds bases at all, refilled byte by byte
      const bool in = p0 + BW <= n_bps + 32;
      const uint32_t *src = reinterpret_cast<const uint32_t *>(gseq + (in ? p0 : 0));
#pragma unroll
      for (int t = 0; t < NDW; ++t) x[t] = src[t];
      if (!in && p0 < n_bps) {
#pragma unroll 1
        for (int t = 0; t < NDW; ++t) {
          uint32_t w = 0;
          for (int bb = 0; bb < 4; ++bb) {
            const uint64_t pos = p0 + 4 * t + bb;
            w |= (uint32_t)(pos < n_bps ? gseq[pos] : (uint8_t)'N') << (8 * bb);
          }
          x[t] = w;
        }
      }
    }
    // ---- classify 4 bases per dword; code streams of the window: W LSB-first (base b at bits 2b..2b+1 of the
    // 128-bit value), V MSB-first (base 0 in the top two bits) --------------------------------------------------
    uint32_t FA[NDW], CA[NDW];
    uint32_t dacc = 0;
    uint32_t W[4] = {0, 0, 0, 0}, V[4] = {0, 0, 0, 0};
#if HG_U2T_HOIST
    if (u2t) {  // needletail normalize: u/U -> T  ('U' ^ 'T' == 1); ONE uniform branch per window
#pragma unroll
      for (int t = 0; t < NDW; ++t) {
        const uint32_t e = (x[t] & 0xDFDFDFDFu) ^ 0x55555555u;
        const uint32_t nz = ((e & 0x7F7F7F7Fu) + 0x7F7F7F7Fu) | e;  // bit 7 set <=> byte != 'U'
        x[t] ^= (~nz & 0x80808080u) >> 7;
      }
    }
#endif
#pragma unroll
    for (int t = 0; t < NDW; ++t) {
#if !HG_U2T_HOIST
      if (u2t) {
        const uint32_t e = (x[t] & 0xDFDFDFDFu) ^ 0x55555555u;
        const uint32_t nz = ((e & 0x7F7F7F7Fu) + 0x7F7F7F7Fu) | e;
        x[t] ^= (~nz & 0x80808080u) >> 7;
      }
#endif
      const uint32_t xv = x[t];
      const uint32_t u = xv & 0xDFDFDFDFu;
      const uint32_t tt = xv ^ (xv >> 1);
      const uint32_t cd = (tt >> 1) & 0x03030303u;  // A,C,G,T -> 0,1,2,3 per byte
      FA[t] = __builtin_amdgcn_perm(0u, 0x54474341u, cd);  // "ACGT"[code]
      CA[t] = __builtin_amdgcn_perm(0u, 0x41434754u, cd);  // "TGCA"[code]
      dacc |= u ^ FA[t];
      const uint32_t pl = __builtin_amdgcn_udot4(cd, 0x40100401u, 0u, false);  // c0 | c1<<2 | c2<<4 | c3<<6
      const uint32_t pm = __builtin_amdgcn_udot4(cd, 0x01041040u, 0u, false);  // c0<<6 | c1<<4 | c2<<2 | c3
      W[t >> 2] |= pl << (8 * (t & 3));
      V[3 - (t >> 2)] |= pm << (24 - 8 * (t & 3));  // V[3] holds bases 0..15, base 0 on top
    }
    // ---- validity: rare path, taken only by waves that see a non-base or the genome end ----
    const int64_t rem64 = (int64_t)n_bps - (int64_t)p0;
    const uint32_t rem = rem64 >= BW ? (uint32_t)BW : (rem64 <= 0 ? 0u : (uint32_t)rem64);
    uint64_t inv = 0;  // bit b set <=> base b of the window cannot be part of a k-mer
    const bool wave_dirty = __any((dacc != 0) | (rem < (uint32_t)BW));  // wave-uniform
    if (wave_dirty) {
      uint32_t iv[2] = {0, 0};
#pragma unroll
      for (int t = 0; t < NDW; ++t) {
        const uint32_t d = (x[t] & 0xDFDFDFDFu) ^ FA[t];
        const uint32_t z = (((d & 0x7F7F7F7Fu) + 0x7F7F7F7Fu) | d) & 0x80808080u;
        const uint32_t nib = (((z >> 7) * 0x01020408u) >> 24) & 0xFu;
        iv[t >> 3] |= nib << (4 * (t & 7));
      }
      inv = mk64(iv[0], iv[1]);
      if (rem < (uint32_t)BW) inv |= (rem == 0) ? ~0ull : (~0ull << rem);
    }
    // ---- LDS image: F[0..56) then R[i] = comp(F[55 - i]); both strands of the k-mer at window position J are
    // ascending byte ranges: forward [J, J+K), reverse [112 - K - J, 112 - J) ---------------------------------
    {
      uint32_t *w1 = reinterpret_cast<uint32_t *>(mywin);
#pragma unroll
      for (int t = 0; t < NDW; ++t) {
        w1[t] = FA[t];
        w1[NDW + t] = __builtin_amdgcn_perm(0u, CA[NDW - 1 - t], 0x00010203u);
      }
    }

#pragma unroll 1
    for (int grp = 0; grp < GROUPS; ++grp) {
      // the slice's code words: bases [12 grp, 12 grp + 32) are the low 64 bits of W and the top 64 bits of V
      const uint64_t Gc = ~mk64(W[0], W[1]);  // complement codes; read LSB-first this IS the reverse strand
      const uint64_t Gm = mk64(V[2], V[3]);
      const uint32_t inv_g = (uint32_t)inv;
      using lds_u8p = __attribute__((address_space(3))) const uint8_t *;
      using lds_u32p = __attribute__((address_space(3))) const uint32_t *;
      const uint32_t my32 = (uint32_t)(uintptr_t)(lds_u8p)mywin;  // the lane's image as a 32-bit LDS address
      const uint32_t winF = my32 + M * grp, winR = my32 - M * grp;

      uint64_t wq[2][NW];
      auto fetch_words = [&](auto jjc, uint64_t *w) __attribute__((always_inline)) {
        constexpr int jj = decltype(jjc)::value;
        uint64_t fv, rv;
        if constexpr ((K & 1) != 0) {
          // odd K: the compare is decided inside the 2K bits, the values only have to be TOP-aligned (kmer_sample_fast)
          fv = Gm << (2 * jj);
          rv = Gc << (2 * (32 - K - jj));
        } else {
          constexpr uint64_t MASK2K = (1ull << (2 * K)) - 1;
          fv = (Gm >> (2 * (32 - K - jj))) & MASK2K;
          rv = (Gc >> (2 * jj)) & MASK2K;
        }
        uint64_t lt;
        uint32_t off;
        uint32_t bsel;
        asm("v_cmp_lt_u64_e64 %0, %1, %2" : "=s"(lt) : "v"(rv), "v"(fv));
        // forward bytes at image offset jj, reverse bytes at OFFR.  Only the byte shifts (low two bits) go through the
        // per-k-mer select; the aligned parts are compile-time: the forward one (CF) is the immediate offset of the
        // reads for BOTH strands, the difference DR sits in the reverse base pointer (three to six distinct values of
        // winR + DR per slice, computed once per slice and shared by the compiler) -- no v_and / v_add per k-mer
        constexpr int OFFR = 2 * BW - K - jj, CF = jj & ~3, DR = (OFFR & ~3) - CF;
        const uint32_t winRD = winR + (uint32_t)DR;
        asm("v_cndmask_b32_e64 %0, %1, %2, %3" : "=v"(off) : "n"(8 * (jj & 3)), "n"(8 * (OFFR & 3)), "s"(lt));  // in bits
        asm("v_cndmask_b32_e64 %0, %1, %2, %3" : "=v"(bsel) : "v"(winF), "v"(winRD), "s"(lt));
        constexpr int NDR = (K + 3 + 3) / 4;  // aligned dwords that cover any K bytes starting at shift 0..3
        const lds_u32p src = (lds_u32p)(uintptr_t)bsel + CF / 4;
        uint32_t t[NDR + 1], d[2 * NW];
#pragma unroll
        for (int m = 0; m < NDR; ++m) t[m] = src[m];
        t[NDR] = 0;
#pragma unroll
        for (int m = 0; m < 2 * NW; ++m) {
          if (m < ND) d[m] = __builtin_amdgcn_alignbit(t[m + 1 < NDR ? m + 1 : NDR], t[m], off);
          else d[m] = 0;
        }
        if constexpr (NB == 1) {
          // a k-mer's last byte lies inside its last aligned dword whatever the shift: one v_bfe_u32 instead of
          // v_alignbit + v_and
          d[ND - 1] = __builtin_amdgcn_ubfe(t[ND - 1], off, 8u);
        } else if constexpr (NB < 4) {
          d[ND - 1] &= (1u << (8 * (NB & 3))) - 1;
        }
#pragma unroll
        for (int m = 0; m < NW; ++m) w[m] = mk64(d[2 * m], d[2 * m + 1]);
      };
      auto run_kmers = [&](auto checkc) __attribute__((always_inline)) {
        constexpr bool CHECK = decltype(checkc)::value;
        static_for(std::make_integer_sequence<int, M>{}, [&](auto jc) {
          constexpr int j = decltype(jc)::value;
          const bool valid = !CHECK || ((inv_g >> j) & MASKK) == 0;
          if constexpr (j == 0) fetch_words(jc, wq[0]);
          if constexpr (j + 1 < M) fetch_words(std::integral_constant<int, j + 1>{}, wq[(j + 1) & 1]);
          const uint64_t h = t1ha2_fixed_w<K>(wq[j & 1], seed);
          // (a 32-bit pre-test of the high dword in front of this 64-bit compare measured +0.8 %)
          if (valid && h < threshold) stage_hit(stage, h, gm, g, hits, cnt);
        });
      };
      if (!wave_dirty) run_kmers(std::false_type{});
      else run_kmers(std::true_type{});

      // slide to the next slice: M bases = 2 M bits of both streams, M bits of the validity mask
      constexpr int SB = 2 * M;
      W[0] = __builtin_amdgcn_alignbit(W[1], W[0], SB), W[1] = __builtin_amdgcn_alignbit(W[2], W[1], SB);
      W[2] = __builtin_amdgcn_alignbit(W[3], W[2], SB), W[3] >>= SB;
      V[3] = __builtin_amdgcn_alignbit(V[3], V[2], 32 - SB), V[2] = __builtin_amdgcn_alignbit(V[2], V[1], 32 - SB);
      V[1] = __builtin_amdgcn_alignbit(V[1], V[0], 32 - SB), V[0] <<= SB;
      inv >>= M;
    }
  }
  flush_hits(stage, gm, g, hits, cnt);
}

// =========================================================================================
// shared-image kernel: compile-time k in [1, 21], both strand modes
// =========================================================================================
// What the grouped kernel still pays per k-mer besides the hash is the extraction of the chosen strand's bytes: the
// k-mer starts at an arbitrary byte of the lane's LDS image, so the six hash dwords are cut out of seven aligned ones
// with a run-time byte shift (5 v_alignbit + v_bfe), and the shift itself has to be selected with the strand (a second
// v_cndmask) -- eight slow-class instructions, 12 % of the issue time.  Here the byte shift is taken out of the inner loop
// altogether: the WORKGROUP keeps one image of its tile in LDS in all four byte phases -- F_phi[i] = bytes
// [4 i + phi, 4 i + phi + 4) of the tile, phi = 0..3 -- and the same for the reverse complement R[q] = comp(F[N - 1 - q]).
// A lane owns M = 12 consecutive starts p = 12 tid + j, so the phase of k-mer j is j & 3 on the forward strand and, with
// N chosen such that (N - K) & 3 == 3, 3 - (j & 3) on the reverse strand: compile-time per j.  Both strands' hash words are
// then whole dwords at [base + imm(j) + 4 m]: ONE v_cndmask picks the base (forward: FB + 12 tid; reverse: one of three
// per-lane constants RB' - 12 tid - 8 (j >> 2)), the immediates imm(j) = S (j & 3) + 4 (j >> 2) are shared by both strands
// (reverse phase psi is stored at S (3 - psi) for that), and the ds_reads deliver the words straight into the register
// pairs the multiplies take -- no VALU instruction touches the bytes.  The images are written once per tile by the lanes
// that load the bases (12 bases + one lookahead dword per lane: every base is classified once, not 32/12 times as in
// kmer_sample_fast or 56/36 as in the grouped kernel); the 2-bit codes for the strand compare and the validity bits go
// through LDS as well (768 + 1 036 bytes).  Two barriers per tile; 28 KB of LDS per workgroup (5 workgroups per CU).

// ---- the k-mer body of kmer_sample_shared in assembly (17 <= K <= 24: three mixup64 stages + final64) -----------------
// The compiler's code for t1ha2_fixed_w spends 10 v_mov and 3 v_add_u32 per hash on moving 32-bit halves into the
// even-aligned register pairs v_mad_u64_u32 / v_lshl_add_u64 take (inline asm operands cannot name the halves of a
// 64-bit operand, so every product's halves travel as separate values and get re-paired), plus nops behind every carry
// that goes through an SGPR pair.  Here all temporaries are fixed physical registers, every result is produced in the
// pair that consumes it, carries go through VCC into a VOP2 v_addc (no SGPR read hazard), and what is left is what the
// ISA forces: the high dword of a product is an ODD register and a 64-bit addend has to start at an EVEN one -- one
// v_mov per 128-bit product (into the pair Z = {x, 0}).  57 vector instructions per hash + compare instead of 64.
//   v[56:61] / v[62:67]  hash words of k-mer j / j+1 (filled by HG_KS_READ one k-mer ahead)
//   Z v[68:69]  H v[70:71]  X v[72:73]  A v[74:75]  T v[76:77]  U v[78:79]  S1 v[80:81]  S2 v[82:83]  R v[84:85]
//   Y v[86:87]  C v[88:89]
#define HG_KS_MUL128(XLO, XHI, PLO, PHI, ADDEND, OUT, OUTHI)                                   \
  "v_mad_u64_u32 v[74:75], %[junk], " XLO ", " PLO ", 0\n\t"                                   \
  "v_mov_b32 v68, v75\n\t"                                                                     \
  "v_mad_u64_u32 v[76:77], %[junk], " XHI ", " PLO ", v[68:69]\n\t"                            \
  "v_mad_u64_u32 v[76:77], vcc, " XLO ", " PHI ", v[76:77]\n\t"                                \
  "v_mad_u64_u32 " OUT ", %[junk], v77, 1, " ADDEND "\n\t"                                     \
  "v_mad_u64_u32 " OUT ", %[junk], " XHI ", " PHI ", " OUT "\n\t"                              \
  "v_addc_co_u32_e32 " OUTHI ", vcc, 0, " OUTHI ", vcc\n\t"
// lo64(x * P) -> {OLO, OHI} (a pair): three chained products and one add into the pair's own high half
#define HG_KS_LO64(XLO, XHI, PLO, PHI, OUT, OLO_UNUSED, OHI)                                   \
  "v_mad_u64_u32 " OUT ", %[junk], " XLO ", " PLO ", 0\n\t"                                    \
  "v_mad_u64_u32 v[76:77], %[junk], " XLO ", " PHI ", 0\n\t"                                   \
  "v_mad_u64_u32 v[76:77], %[junk], " XHI ", " PLO ", v[76:77]\n\t"                            \
  "v_add_u32_e32 " OHI ", " OHI ", v76\n\t"
// request the words of one k-mer: cases by (dwords, bytes in the last dword) = K 17 (5,1), 18 (5,2), 19 (5,3), 20 (5,4), 21 (6,1)
// (four dword reads: a ds_read_b64 at an address that is a multiple of 4 but not of 8 is as slow as a byte-aligned one --
// measured 18.8 ms against 8.5 for the whole kernel -- and ds_read2_b32's 8-bit offsets cannot hold the phase image's)
#define HG_KS_READ_HEAD(R0, R1, R2, R3)                                                                              \
  "ds_read_b32 v" R0 ", %[base] offset:%[o0]\n\tds_read_b32 v" R1 ", %[base] offset:%[o0] + 4\n\t"                   \
  "ds_read_b32 v" R2 ", %[base] offset:%[o1]\n\tds_read_b32 v" R3 ", %[base] offset:%[o1] + 4\n\t"
#define HG_KS_READ_TEXT_51(R0, R1, R2, R3, R4, R5) HG_KS_READ_HEAD(R0, R1, R2, R3) "ds_read_u8 v" R4 ", %[base] offset:%[o2]\n\tv_mov_b32 v" R5 ", 0"
#define HG_KS_READ_TEXT_52(R0, R1, R2, R3, R4, R5) HG_KS_READ_HEAD(R0, R1, R2, R3) "ds_read_u16 v" R4 ", %[base] offset:%[o2]\n\tv_mov_b32 v" R5 ", 0"
#define HG_KS_READ_TEXT_53(R0, R1, R2, R3, R4, R5) HG_KS_READ_HEAD(R0, R1, R2, R3) "ds_read_b32 v" R4 ", %[base] offset:%[o2]\n\tv_mov_b32 v" R5 ", 0"
#define HG_KS_READ_TEXT_54(R0, R1, R2, R3, R4, R5) HG_KS_READ_TEXT_53(R0, R1, R2, R3, R4, R5)
#define HG_KS_READ_TEXT_61(R0, R1, R2, R3, R4, R5) HG_KS_READ_HEAD(R0, R1, R2, R3) "ds_read_b32 v" R4 ", %[base] offset:%[o2]\n\tds_read_u8 v" R5 ", %[base] offset:%[o3]"
// wait for the words requested one hash ago; K = 19: the last dword carries a byte of the next base
#define HG_KS_WAIT_TEXT_51(R4) "s_waitcnt lgkmcnt(0)\n\t"
#define HG_KS_WAIT_TEXT_52(R4) "s_waitcnt lgkmcnt(0)\n\t"
#define HG_KS_WAIT_TEXT_53(R4) "s_waitcnt lgkmcnt(0)\n\tv_and_b32 v" R4 ", 0xffffff, v" R4 "\n\t"
#define HG_KS_WAIT_TEXT_54(R4) "s_waitcnt lgkmcnt(0)\n\t"
#define HG_KS_WAIT_TEXT_61(R4) "s_waitcnt lgkmcnt(0)\n\t"
#define HG_KS_HASH_TEXT(W0, W1, W2)                                                                                  \
  /* mixup64<P3>(b, a, w0): x = seed + w0; b = K ^ lo; a = seed + hi */                                              \
  "v_lshl_add_u64 v[72:73], " W0 ", 0, %[seed]\n\t"                                                                  \
  HG_KS_MUL128("v72", "v73", "%[p3l]", "%[p3h]", "%[seed]", "v[78:79]", "v79")                                       \
  "v_xor_b32_e32 v80, %[kk], v74\n\t"                                                                                \
  "v_mov_b32 v81, v76\n\t"                                                                                           \
  /* mixup64<P2>(a, b, w1): x = b + w1; a ^= lo; b += hi   (a = U, b = S1 -> a' = S2, b' = R) */                     \
  "v_lshl_add_u64 v[72:73], v[80:81], 0, " W1 "\n\t"                                                                 \
  HG_KS_MUL128("v72", "v73", "%[p2l]", "%[p2h]", "v[80:81]", "v[84:85]", "v85")                                      \
  "v_xor_b32_e32 v82, v78, v74\n\t"                                                                                  \
  "v_xor_b32_e32 v83, v79, v76\n\t"                                                                                  \
  /* mixup64<P1>(b, a, w2): x = a + w2; b ^= lo; a += hi   (a = S2, b = R -> b'' = S1, a'' = U) */                   \
  "v_lshl_add_u64 v[72:73], v[82:83], 0, " W2 "\n\t"                                                                 \
  HG_KS_MUL128("v72", "v73", "%[p1l]", "%[p1h]", "v[82:83]", "v[78:79]", "v79")                                      \
  "v_xor_b32_e32 v80, v84, v74\n\t"                                                                                  \
  "v_xor_b32_e32 v81, v85, v76\n\t"                                                                                  \
  /* final64(a = U, b = S1): x = (a + rot64(b, 41)) * P0, y = (rot64(a, 23) + b) * P6 (low halves), z = x ^ y */      \
  "v_alignbit_b32 v86, v80, v81, 9\n\t"                                                                              \
  "v_alignbit_b32 v87, v81, v80, 9\n\t"                                                                              \
  "v_lshl_add_u64 v[72:73], v[86:87], 0, v[78:79]\n\t"                                                               \
  "v_alignbit_b32 v86, v79, v78, 23\n\t"                                                                             \
  "v_alignbit_b32 v87, v78, v79, 23\n\t"                                                                             \
  "v_lshl_add_u64 v[88:89], v[86:87], 0, v[80:81]\n\t"                                                               \
  HG_KS_LO64("v72", "v73", "%[p0l]", "%[p0h]", "v[74:75]", "v74", "v75")                                             \
  HG_KS_LO64("v88", "v89", "%[p6l]", "%[p6h]", "v[84:85]", "v84", "v85")                                             \
  "v_xor_b32_e32 v72, v74, v84\n\t"                                                                                  \
  "v_xor_b32_e32 v73, v75, v85\n\t"                                                                                  \
  /* mux64(z, P5): lo ^ hi of the 128-bit product */                                                                 \
  "v_mad_u64_u32 v[74:75], %[junk], v72, %[p5l], 0\n\t"                                                              \
  "v_mov_b32 v68, v75\n\t"                                                                                           \
  "v_mad_u64_u32 v[76:77], %[junk], v73, %[p5l], v[68:69]\n\t"                                                       \
  "v_mad_u64_u32 v[76:77], vcc, v72, %[p5h], v[76:77]\n\t"                                                           \
  "v_mov_b32 v68, v77\n\t"                                                                                           \
  "v_mad_u64_u32 v[78:79], %[junk], v73, %[p5h], v[68:69]\n\t"                                                       \
  "v_addc_co_u32_e32 v79, vcc, 0, v79, vcc\n\t"                                                                      \
  "v_xor_b32_e32 v70, v74, v78\n\t"                                                                                  \
  "v_xor_b32_e32 v71, v76, v79\n\t"                                                                                  \
  "v_cmp_gt_u64_e64 %[mask], %[thr], v[70:71]"
#define HG_KS_HASH_INPUTS                                                                                            \
  [seed] "s"(seed), [thr] "s"(threshold), [kk] "n"(K), [p0l] "s"((uint32_t)P0), [p0h] "s"((uint32_t)(P0 >> 32)),     \
      [p1l] "s"((uint32_t)P1), [p1h] "s"((uint32_t)(P1 >> 32)), [p2l] "s"((uint32_t)P2), [p2h] "s"((uint32_t)(P2 >> 32)), \
      [p3l] "s"((uint32_t)P3), [p3h] "s"((uint32_t)(P3 >> 32)), [p5l] "s"((uint32_t)P5), [p5h] "s"((uint32_t)(P5 >> 32)), \
      [p6l] "s"((uint32_t)P6), [p6h] "s"((uint32_t)(P6 >> 32))
#define HG_KS_HASH_CLOBBERS                                                                                          \
  "vcc", "v72", "v73", "v74", "v75", "v76", "v77", "v78", "v79", "v80", "v81", "v82", "v83", "v84",   \
      "v85", "v86", "v87", "v88", "v89"

template <int K>
struct GeoS {
  static constexpr int M = 12;                      // starts per lane and tile (multiple of 4: the phase of start j is j & 3)
  static constexpr int DW = M / 4;
  // Every lane stages one unit of M bases; the k-mers of the last lanes' windows would need bases behind the staged
  // area, so the last LOOK_UNITS lanes hash nothing and the tile advances by (WG - LOOK_UNITS) * M starts: 0.8 % of the
  // hashing slots of one wave idle -- against a second, three-lane staging pass on the way to the barrier (+2 %)
  static constexpr int LOOK_UNITS = (32 - M + M - 1) / M;  // the last hashing lane's 32-base window ends inside the staged area: 2 units at M = 12
  static constexpr int TILE = (WG - LOOK_UNITS) * M;  // 3 048 starts
  static constexpr int TILES = 9;
  static constexpr int ITEM = TILE * TILES;         // 27 432 starts per work item
  static constexpr int UNITS = WG;
  static constexpr int NB_T = UNITS * M;            // staged bases (3 072)
  static constexpr int N_R = NB_T + ((K + 3 - NB_T) & 3);  // length the reverse strand is indexed in: (N_R - K) & 3 == 3
  static constexpr int S = 4 * (NB_T / 4 + 3);      // bytes per phase image
  static constexpr int ND = (K + 3) / 4, NB = K - 4 * (ND - 1), NW = (K + 7) / 8;
  static_assert(M + K - 1 <= 32, "a lane's k-mers live in a 32-base code window");
  static_assert(((N_R - K) & 3) == 3, "reverse phase of start j is 3 - (j & 3)");
};

template <int K, bool CANON>
__global__ __launch_bounds__(WG) void kmer_sample_shared(
    const uint8_t *__restrict__ seq, const hg_genome_meta *__restrict__ meta,
    const uint32_t *__restrict__ item_genome, uint64_t threshold, uint64_t seed, uint32_t u2t,
    uint64_t *__restrict__ hits, uint32_t *__restrict__ cnt) {
  using G = GeoS<K>;
  constexpr int M = G::M, DW = G::DW, S = G::S, ND = G::ND, NB = G::NB, NW = G::NW, N_R = G::N_R;
  constexpr uint32_t MASKK = (1u << K) - 1;

  const uint32_t item = blockIdx.x, tid = threadIdx.x;
  const uint32_t g = item_genome[item];
  const hg_genome_meta gm = meta[g];
  const uint64_t n_bps = gm.n_bps;
  if (n_bps < (uint64_t)K) return;
  const uint64_t n_starts = n_bps - K + 1;
  const uint8_t *__restrict__ gseq = seq + gm.seq_off;
  const uint64_t item_start = (uint64_t)(item - gm.item_first) * G::ITEM;

  __shared__ HitStage stage;
  __shared__ __attribute__((aligned(16))) uint32_t s_f[4 * S / 4];                 // forward phase images
  __shared__ __attribute__((aligned(16))) uint32_t s_r[CANON ? 4 * S / 4 : 4];     // reverse phase images (psi at S (3 - psi))
  __shared__ __attribute__((aligned(16))) uint8_t s_code[CANON ? G::NB_T / 4 + 16 : 16];  // 2-bit codes, base b at bits 2b..2b+1
  __shared__ uint32_t s_val[G::UNITS + 2];   // per unit of M bases: bit b set <=> base b cannot be part of a k-mer
  __shared__ uint32_t s_dirty[2];            // "some base of this tile is not ACGT / lies behind the genome end", by tile parity
  if (tid == 0) stage.n = 0, s_dirty[0] = 0, s_dirty[1] = 0;
  if (tid < 2) s_val[G::UNITS + tid] = 0;
  __syncthreads();

  using lds_u8p = __attribute__((address_space(3))) const uint8_t *;
  using lds_u16p = __attribute__((address_space(3))) const uint16_t *;
  using lds_u32p = __attribute__((address_space(3))) const uint32_t *;
  // per-lane read bases (they do not depend on the tile)
  const uint32_t aF = (uint32_t)(uintptr_t)(lds_u8p)(reinterpret_cast<const uint8_t *>(s_f)) + M * tid;
  uint32_t aR[M / 4];
#pragma unroll
  for (int gq = 0; gq < M / 4; ++gq)
    aR[gq] = (uint32_t)(uintptr_t)(lds_u8p)(reinterpret_cast<const uint8_t *>(s_r)) + 4u * ((uint32_t)(N_R - K) >> 2) - M * tid - 8u * gq;

  // stage unit u of the tile: bases [M u, M u + M) at genome position P, one lookahead dword
  auto stage_unit = [&](uint32_t u, uint64_t tile_start, uint32_t par) __attribute__((always_inline)) {
    const uint64_t P = tile_start + (uint64_t)u * M;
    uint32_t x[DW + 1];
    const int64_t rem64 = (int64_t)n_bps - (int64_t)P;  // bases of the genome from P on
    if (rem64 + 32 >= 4 * (DW + 1)) {  // the caller provides 32 readable bytes behind every genome
      const uint32_t *src = reinterpret_cast<const uint32_t *>(gseq + P);
#pragma unroll
      for (int t = 0; t <= DW; ++t) x[t] = src[t];
    } else {
#pragma unroll
      for (int t = 0; t <= DW; ++t) {
        uint32_t w = 0;
        for (int bb = 0; bb < 4; ++bb) {
          const int64_t o = 4 * t + bb;
          w |= (uint32_t)(o < rem64 ? gseq[P + o] : (uint8_t)'N') << (8 * bb);
        }
        x[t] = w;
      }
    }
    if (u2t) {  // needletail normalize: u/U -> T  ('U' ^ 'T' == 1); one uniform branch
#pragma unroll
      for (int t = 0; t <= DW; ++t) {
        const uint32_t e = (x[t] & 0xDFDFDFDFu) ^ 0x55555555u;
        const uint32_t nz = ((e & 0x7F7F7F7Fu) + 0x7F7F7F7Fu) | e;  // bit 7 set <=> byte != 'U'
        x[t] ^= (~nz & 0x80808080u) >> 7;
      }
    }
    uint32_t FA[DW + 1], CA[DW + 1];
    uint32_t dacc = 0, codes = 0;
#pragma unroll
    for (int t = 0; t <= DW; ++t) {
      const uint32_t xv = x[t];
      const uint32_t tt = xv ^ (xv >> 1);
      const uint32_t cd = (tt >> 1) & 0x03030303u;            // A,C,G,T -> 0,1,2,3 per byte
      FA[t] = __builtin_amdgcn_perm(0u, 0x54474341u, cd);     // "ACGT"[code]
      if (CANON) CA[t] = __builtin_amdgcn_perm(0u, 0x41434754u, cd);  // "TGCA"[code]
      if (t < DW) {
        dacc |= (xv & 0xDFDFDFDFu) ^ FA[t];
        if (CANON) codes |= __builtin_amdgcn_udot4(cd, 0x40100401u, 0u, false) << (8 * t);  // c0 | c1<<2 | c2<<4 | c3<<6
      }
    }
    // validity of the unit's own M bases
    uint32_t inv = 0;
    const bool dirty = dacc != 0 || rem64 < M;
    if (__any(dirty)) {
#pragma unroll
      for (int t = 0; t < DW; ++t) {
        const uint32_t d = (x[t] & 0xDFDFDFDFu) ^ FA[t];
        const uint32_t z = (((d & 0x7F7F7F7Fu) + 0x7F7F7F7Fu) | d) & 0x80808080u;
        inv |= ((((z >> 7) * 0x01020408u) >> 24) & 0xFu) << (4 * t);
      }
      if (rem64 < M) inv |= rem64 <= 0 ? ~0u : (~0u << (uint32_t)rem64);
      inv &= (1u << M) - 1;
      if ((threadIdx.x & 63) == 0 || dirty) s_dirty[par] = 1u;  // (same value from every writer)
    }
    s_val[u] = inv;
    // forward phase images: F_phi[DW u + t] = bytes [4 t + phi, 4 t + phi + 4) of the unit (+ lookahead)
    uint32_t *const f0 = s_f + DW * u;
#pragma unroll
    for (int t = 0; t < DW; ++t) {
      f0[t] = FA[t];
      f0[S / 4 + t] = __builtin_amdgcn_alignbyte(FA[t + 1], FA[t], 1);
      f0[2 * (S / 4) + t] = __builtin_amdgcn_alignbyte(FA[t + 1], FA[t], 2);
      f0[3 * (S / 4) + t] = __builtin_amdgcn_alignbyte(FA[t + 1], FA[t], 3);
    }
    if constexpr (CANON) {
      // 2-bit codes: M / 4 bytes at byte M u / 4
#pragma unroll
      for (int t = 0; t < DW; ++t) s_code[DW * u + t] = (uint8_t)(codes >> (8 * t));
      // reverse phase images: the forward bytes [s, s + 4), s = 4 i + phi', complemented and byte-reversed, are dword
      // j = JB(phi') - i of reverse phase psi = (N_R - phi') & 3, which lives at S (3 - psi)
#pragma unroll
      for (int ph = 0; ph < 4; ++ph) {
        constexpr int dummy = 0;
        (void)dummy;
        const int psi = (N_R - ph) & 3, JB = (N_R - 4 - psi - ph) / 4;
        const uint32_t sel = (uint32_t)(ph + 3) | ((uint32_t)(ph + 2) << 8) | ((uint32_t)(ph + 1) << 16) | ((uint32_t)ph << 24);
        uint32_t *const r0 = s_r + (3 - psi) * (S / 4) + JB - (int)(DW * u);
#pragma unroll
        for (int t = 0; t < DW; ++t) r0[-t] = __builtin_amdgcn_perm(CA[t + 1], CA[t], sel);
        // i = -1: the bytes in front of the tile do not exist; what follows them (the tile's first ph bases) does
        if (ph > 0 && u == 0) r0[1] = __builtin_amdgcn_perm(CA[0], 0u, sel);
      }
    }
  };

#pragma unroll 1
  for (int tile = 0; tile < G::TILES; ++tile) {
    const uint64_t tile_start = item_start + (uint64_t)tile * G::TILE;
    if (tile_start >= n_starts) break;  // uniform
    const uint32_t par = (uint32_t)tile & 1u;
    stage_unit(tid, tile_start, par);
#ifndef HG_KS_EXP
#define HG_KS_EXP 0
#endif
    if (!(HG_KS_EXP & 1)) __syncthreads();

    const bool tile_dirty = s_dirty[par] != 0u;  // workgroup-uniform
    uint32_t inv32 = 0;
    if (tile_dirty) inv32 = s_val[tid] | (s_val[tid + 1] << M) | (s_val[tid + 2] << (2 * M));
    uint64_t Gm = 0, Gc = 0;
    if constexpr (CANON) {
      // the lane's 32-base code window: 64 bits from bit 2 M tid = byte (M / 4) tid of the code image
      const uint32_t bo = (uint32_t)DW * tid;
      const uint32_t *cw = reinterpret_cast<const uint32_t *>(s_code) + (bo >> 2);
      const uint32_t c0 = cw[0], c1 = cw[1], c2 = cw[2], sh = 8u * (bo & 3u);
      const uint32_t wl = __builtin_amdgcn_alignbit(c1, c0, sh), wh = __builtin_amdgcn_alignbit(c2, c1, sh);
      auto pairrev = [](uint32_t v) {
        const uint32_t br = __builtin_bitreverse32(v);
        return ((br >> 1) & 0x55555555u) | ((br & 0x55555555u) << 1);
      };
      Gc = ~mk64(wl, wh);                       // complement codes; read LSB-first this IS the reverse strand
      Gm = mk64(pairrev(wh), pairrev(wl));      // MSB-first copy: base 0 in the top two bits
    }

    auto fetch_words = [&](auto jjc, uint64_t *w) __attribute__((always_inline)) {
      constexpr int jj = decltype(jjc)::value;
      uint32_t base = aF;
      if constexpr (CANON) {
        uint64_t fv, rv;
        if constexpr ((K & 1) != 0) {
          // odd K: the compare is decided inside the 2K bits, the values only have to be TOP-aligned
          fv = Gm << (2 * jj);
          rv = Gc << (2 * (32 - K - jj));
        } else {
          constexpr uint64_t MASK2K = (1ull << (2 * K)) - 1;
          fv = (Gm >> (2 * (32 - K - jj))) & MASK2K;
          rv = (Gc >> (2 * jj)) & MASK2K;
        }
        uint64_t lt;
        asm("v_cmp_lt_u64_e64 %0, %1, %2" : "=s"(lt) : "v"(rv), "v"(fv));
        asm("v_cndmask_b32_e64 %0, %1, %2, %3" : "=v"(base) : "v"(aF), "v"(aR[jj >> 2]), "s"(lt));
      }
      constexpr int IMM = S * (jj & 3) + 4 * (jj >> 2);
      const lds_u32p src = (lds_u32p)(uintptr_t)(base + (uint32_t)IMM);
      uint32_t d[2 * NW];
#pragma unroll
      for (int m = 0; m < 2 * NW; ++m) {
        if (m < ND - 1 || (m == ND - 1 && NB == 4)) d[m] = src[m];
        else if (m == ND - 1 && NB == 1) d[m] = *(lds_u8p)(uintptr_t)(base + (uint32_t)(IMM + 4 * m));
        else if (m == ND - 1 && NB == 2) d[m] = *(lds_u16p)(uintptr_t)(base + (uint32_t)(IMM + 4 * m));
        else if (m == ND - 1) d[m] = src[m] & 0xFFFFFFu;
        else d[m] = 0;
      }
#pragma unroll
      for (int m = 0; m < NW; ++m) w[m] = mk64(d[2 * m], d[2 * m + 1]);
    };
    uint64_t wq[2][NW];
    const bool hashing = tid < (uint32_t)(WG - G::LOOK_UNITS);  // (the last lanes' windows leave the staged area)
#ifndef HG_KS_ASM
#define HG_KS_ASM 1  /* 17 <= K <= 24: the k-mer body in assembly (0: the compiler's code for the same arithmetic, A/B) */
#endif
    constexpr bool ASM_BODY = HG_KS_ASM && NW == 3 && (ND == 5 || (ND == 6 && NB == 1));  // K = 17..21
    // the strand's base address for k-mer jj (ASM_BODY; the compiler path has it inside fetch_words)
    auto strand_base = [&](auto jjc) __attribute__((always_inline)) -> uint32_t {
      constexpr int jj = decltype(jjc)::value;
      uint32_t base = aF;
      if constexpr (CANON) {
        uint64_t fv, rv;
        if constexpr ((K & 1) != 0) {
          fv = Gm << (2 * jj);
          rv = Gc << (2 * (32 - K - jj));
        } else {
          constexpr uint64_t MASK2K = (1ull << (2 * K)) - 1;
          fv = (Gm >> (2 * (32 - K - jj))) & MASK2K;
          rv = (Gc >> (2 * jj)) & MASK2K;
        }
        uint64_t lt;
        asm("v_cmp_lt_u64_e64 %0, %1, %2" : "=s"(lt) : "v"(rv), "v"(fv));
        asm("v_cndmask_b32_e64 %0, %1, %2, %3" : "=v"(base) : "v"(aF), "v"(aR[jj >> 2]), "s"(lt));
      }
      return base;
    };
    // One asm statement per k-mer: wait for this k-mer's words (requested a whole hash earlier), request the next
    // k-mer's into the other buffer (whole dwords of the chosen strand's phase image: two 8-byte reads at 4-byte aligned
    // addresses, one dword and the last 1..4 bytes), hash, compare with the threshold.  The buffers are bound to fixed
    // registers on both sides, so the compiler sees ordinary values and never copies them.
    uint64_t w0a = 0, w0b = 0, w0c = 0, w1a = 0, w1b = 0, w1c = 0;  // parity 0: v[56:61], parity 1: v[62:67]
    uint64_t hmask = 0, hjunk = 0, hval = 0;
    uint64_t zpair = 0;  // Z = {x, 0}: the zero-extension pair of the products' high dwords; every asm statement rewrites
                         // its low half only, so the zero in the high half is carried from k-mer to k-mer as a value
    constexpr int KCASE = 10 * ND + NB;
#define HG_KS_FIRST(C)                                                                                                \
  if constexpr (KCASE == C)                                                                                           \
    asm volatile(HG_KS_READ_TEXT_##C("56", "57", "58", "59", "60", "61")                                             \
                 : "={v[56:57]}"(w0a), "={v[58:59]}"(w0b), "={v[60:61]}"(w0c)                                         \
                 : [base] "v"(base), [o0] "n"(IMM), [o1] "n"(IMM + 8), [o2] "n"(IMM + 16), [o3] "n"(IMM + 20));
#define HG_KS_EVEN(C)                                                                                                 \
  if constexpr (KCASE == C)                                                                                           \
    asm volatile(HG_KS_WAIT_TEXT_##C("60") HG_KS_READ_TEXT_##C("62", "63", "64", "65", "66", "67") "\n\t"            \
                 HG_KS_HASH_TEXT("v[56:57]", "v[58:59]", "v[60:61]")                                                  \
                 : [mask] "=s"(hmask), [junk] "=&s"(hjunk), "={v[70:71]}"(hval), "={v[62:63]}"(w1a), "={v[64:65]}"(w1b), \
                   "={v[66:67]}"(w1c), "={v[68:69]}"(zpair)                                                           \
                 : HG_KS_HASH_INPUTS, "{v[68:69]}"(zpair), [base] "v"(base), [o0] "n"(IMM), [o1] "n"(IMM + 8), [o2] "n"(IMM + 16),         \
                   [o3] "n"(IMM + 20), "{v[56:57]}"(w0a), "{v[58:59]}"(w0b), "{v[60:61]}"(w0c)                        \
                 : HG_KS_HASH_CLOBBERS);
#define HG_KS_ODD(C)                                                                                                  \
  if constexpr (KCASE == C)                                                                                           \
    asm volatile(HG_KS_WAIT_TEXT_##C("66") HG_KS_READ_TEXT_##C("56", "57", "58", "59", "60", "61") "\n\t"            \
                 HG_KS_HASH_TEXT("v[62:63]", "v[64:65]", "v[66:67]")                                                  \
                 : [mask] "=s"(hmask), [junk] "=&s"(hjunk), "={v[70:71]}"(hval), "={v[56:57]}"(w0a), "={v[58:59]}"(w0b), \
                   "={v[60:61]}"(w0c), "={v[68:69]}"(zpair)                                                           \
                 : HG_KS_HASH_INPUTS, "{v[68:69]}"(zpair), [base] "v"(base), [o0] "n"(IMM), [o1] "n"(IMM + 8), [o2] "n"(IMM + 16),         \
                   [o3] "n"(IMM + 20), "{v[62:63]}"(w1a), "{v[64:65]}"(w1b), "{v[66:67]}"(w1c)                        \
                 : HG_KS_HASH_CLOBBERS);
#define HG_KS_LAST(C)                                                                                                 \
  if constexpr (KCASE == C)                                                                                           \
    asm volatile(HG_KS_WAIT_TEXT_##C("66") HG_KS_HASH_TEXT("v[62:63]", "v[64:65]", "v[66:67]")                        \
                 : [mask] "=s"(hmask), [junk] "=&s"(hjunk), "={v[70:71]}"(hval), "={v[68:69]}"(zpair)                 \
                 : HG_KS_HASH_INPUTS, "{v[68:69]}"(zpair), "{v[62:63]}"(w1a), "{v[64:65]}"(w1b), "{v[66:67]}"(w1c)    \
                 : HG_KS_HASH_CLOBBERS);
#define HG_KS_ALL_CASES(X) X(51) X(52) X(53) X(54) X(61)
    auto kmers_asm = [&](auto checkc) __attribute__((always_inline)) {
      constexpr bool CHECK = decltype(checkc)::value;
      if constexpr (ASM_BODY) {
        {
          const uint32_t base = strand_base(std::integral_constant<int, 0>{});
          constexpr int IMM = 0;
          HG_KS_ALL_CASES(HG_KS_FIRST)
        }
        static_for(std::make_integer_sequence<int, M>{}, [&](auto jc) {
          constexpr int j = decltype(jc)::value;
          const bool valid = !CHECK || ((inv32 >> j) & MASKK) == 0;
          if constexpr (j + 1 < M) {
            const uint32_t base = strand_base(std::integral_constant<int, j + 1>{});
            constexpr int IMM = S * ((j + 1) & 3) + 4 * ((j + 1) >> 2);
            if constexpr ((j & 1) == 0) {
              HG_KS_ALL_CASES(HG_KS_EVEN)
            } else {
              HG_KS_ALL_CASES(HG_KS_ODD)
            }
          } else {
            static_assert((M & 1) == 0, "the last k-mer's words are in the parity-1 buffer");
            HG_KS_ALL_CASES(HG_KS_LAST)
          }
          if (hmask != 0) {  // wave-uniform: some lane's hash is below the threshold (1 k-mer in `scaled`)
            uint32_t below;
            asm volatile("v_cndmask_b32_e64 %0, 0, 1, %1" : "=v"(below) : "s"(hmask));
            if (below && valid && hashing) stage_hit(stage, hval, gm, g, hits, cnt);
          }
        });
      }
    };
#undef HG_KS_FIRST
#undef HG_KS_EVEN
#undef HG_KS_ODD
#undef HG_KS_LAST
    auto run_kmers = [&](auto checkc) __attribute__((always_inline)) {
      constexpr bool CHECK = decltype(checkc)::value;
      static_for(std::make_integer_sequence<int, M>{}, [&](auto jc) {
        constexpr int j = decltype(jc)::value;
        const bool valid = !CHECK || ((inv32 >> j) & MASKK) == 0;
        if constexpr (j == 0) fetch_words(jc, wq[0]);
        if constexpr (j + 1 < M) fetch_words(std::integral_constant<int, j + 1>{}, wq[(j + 1) & 1]);
        const uint64_t h = t1ha2_fixed_w<K>(wq[j & 1], seed);
        if (valid && hashing && h < threshold) stage_hit(stage, h, gm, g, hits, cnt);
      });
    };
    if constexpr (ASM_BODY) {
      if (!tile_dirty) kmers_asm(std::false_type{});
      else kmers_asm(std::true_type{});
    } else {
      if (!tile_dirty) run_kmers(std::false_type{});
      else run_kmers(std::true_type{});
    }
    if (!(HG_KS_EXP & 1)) __syncthreads();  // every read of the images is done: the next tile may overwrite them
    if (tid == 0) s_dirty[par] = 0u;  // (raised again in two tiles' time at the earliest, behind the next tile's barriers)
  }
  flush_hits(stage, gm, g, hits, cnt);
}

// =========================================================================================
// fast kernel, 64-base window: compile-time k in [22, 32]
// =========================================================================================
// Same scheme as kmer_sample_fast with a 64-base register window per lane (16 dwords, 32 k-mer starts,
// 32-byte lane stride); the 2-bit streams are 128 bits wide, k-mer values are still <= 64 bits.
template <int K>
struct Geo64 {
  static constexpr int M = 32;
  static constexpr int ND = (K + 3) / 4;
  static constexpr int NB = K - 4 * (ND - 1);
  static constexpr int TILE = WG * M;
  static constexpr int ITEM = TILE * TILES_PER_ITEM64;
};

template <int K, bool CANON>
__global__ __launch_bounds__(WG) void kmer_sample_fast64(
    const uint8_t *__restrict__ seq, const hg_genome_meta *__restrict__ meta,
    const uint32_t *__restrict__ item_genome, uint64_t threshold, uint64_t seed, uint32_t u2t,
    uint64_t *__restrict__ hits, uint32_t *__restrict__ cnt) {
  using G = Geo64<K>;
  constexpr int M = G::M, ND = G::ND, NB = G::NB;
  constexpr uint64_t MASK2K = (K == 32) ? ~0ull : ((1ull << (2 * K)) - 1);
  constexpr uint64_t MASKK = (1ull << K) - 1;

  const uint32_t item = blockIdx.x;
  const uint32_t g = item_genome[item];
  const hg_genome_meta gm = meta[g];
  const uint64_t n_bps = gm.n_bps;
  if (n_bps < (uint64_t)K) return;
  const uint64_t n_starts = n_bps - K + 1;
  const uint8_t *__restrict__ gseq = seq + gm.seq_off;
  const uint64_t item_start = (uint64_t)(item - gm.item_first) * G::ITEM;
  __shared__ HitStage stage;
  if (threadIdx.x == 0) stage.n = 0;
  __syncthreads();

#pragma unroll 1
  for (int tile = 0; tile < TILES_PER_ITEM64; ++tile) {
    const uint64_t tile_start = item_start + (uint64_t)tile * G::TILE;
    if (tile_start >= n_starts) break;  // uniform
    const uint64_t p0 = tile_start + (uint64_t)threadIdx.x * M;
    uint32_t x[16];
    {
      // reads 64 bytes; the last lanes of a genome may run past its end by up to 63 bytes, so the
      // window start is clamped to stay inside [0, n_bps + 32 - 64] ... lanes whose window would
      // cross the slack are redirected to the genome start and masked out below
      const bool in = p0 + 64 <= n_bps + 32;
      const uint32_t *src = reinterpret_cast<const uint32_t *>(gseq + (in ? p0 : 0));
#pragma unroll
      for (int t = 0; t < 16; ++t) x[t] = src[t];
      if (!in && p0 < n_bps) {  // rare tail lanes: byte-wise, bounded by the genome end
#pragma unroll 1
        for (int t = 0; t < 16; ++t) {
          uint32_t w = 0;
          for (int bb = 0; bb < 4; ++bb) {
            const uint64_t pos = p0 + 4 * t + bb;
            w |= (uint32_t)(pos < n_bps ? gseq[pos] : (uint8_t)'N') << (8 * bb);
          }
          x[t] = w;
        }
      }
    }
    uint32_t FA[16], CA[16];
    uint32_t dacc = 0;
    uint32_t Gw[4] = {0, 0, 0, 0};  // 2-bit codes, base b at bits [2b, 2b+1] of the 128-bit value
#pragma unroll
    for (int t = 0; t < 16; ++t) {
      uint32_t xv = x[t];
      if (u2t) {
        uint32_t e = (xv & 0xDFDFDFDFu) ^ 0x55555555u;
        uint32_t nz = ((e & 0x7F7F7F7Fu) + 0x7F7F7F7Fu) | e;
        xv ^= (~nz & 0x80808080u) >> 7;
        x[t] = xv;
      }
      uint32_t u = xv & 0xDFDFDFDFu;
      uint32_t tt = xv ^ (xv >> 1);
      uint32_t cd = (tt >> 1) & 0x03030303u;
      FA[t] = __builtin_amdgcn_perm(0u, 0x54474341u, cd);
      CA[t] = __builtin_amdgcn_perm(0u, 0x41434754u, cd);
      dacc |= u ^ FA[t];
      const uint32_t p = __builtin_amdgcn_udot4(cd, 0x40100401u, 0u, false);
      Gw[t >> 2] |= p << (8 * (t & 3));
    }
    auto pairrev = [](uint32_t v) {
      uint32_t br = __builtin_bitreverse32(v);
      return ((br >> 1) & 0x55555555u) | ((br & 0x55555555u) << 1);
    };
    // LSB-first 128-bit code stream [Gl1:Gl0], its complement, and the MSB-first copy [Gm1:Gm0]
    const uint64_t Gl0 = mk64(Gw[0], Gw[1]), Gl1 = mk64(Gw[2], Gw[3]);
    const uint64_t Gc0 = ~Gl0, Gc1 = ~Gl1;
    const uint64_t Gm1 = mk64(pairrev(Gw[1]), pairrev(Gw[0]));  // bases 0..31, base 0 on top
    const uint64_t Gm0 = mk64(pairrev(Gw[3]), pairrev(Gw[2]));  // bases 32..63

    const int64_t rem64 = (int64_t)n_bps - (int64_t)p0;
    const uint32_t rem = rem64 >= 64 ? 64u : (rem64 <= 0 ? 0u : (uint32_t)rem64);
    uint64_t inv = 0;
    if (__any((dacc != 0) | (rem < 64))) {
      uint32_t iv[2] = {0, 0};
#pragma unroll
      for (int t = 0; t < 16; ++t) {
        uint32_t d = (x[t] & 0xDFDFDFDFu) ^ FA[t];
        uint32_t z = (((d & 0x7F7F7F7Fu) + 0x7F7F7F7Fu) | d) & 0x80808080u;
        uint32_t nib = (((z >> 7) * 0x01020408u) >> 24) & 0xFu;
        iv[t >> 3] |= nib << (4 * (t & 7));
      }
      inv = mk64(iv[0], iv[1]);
      if (rem < 64) inv |= (rem == 0) ? ~0ull : (~0ull << rem);
    }

    static_for(std::make_integer_sequence<int, M>{}, [&](auto jc) {
      constexpr int j = decltype(jc)::value;
      constexpr int q = j >> 2, r = j & 3;
      const bool valid = ((inv >> j) & MASKK) == 0;
      uint32_t rc_mask = 0;
      if (CANON) {
        constexpr int sf = 2 * (64 - K - j);  // forward value: bits [sf, sf+2K) of [Gm1:Gm0]
        uint64_t fv;
        if constexpr (sf >= 64) fv = Gm1 >> (sf - 64);
        else if constexpr (sf == 0) fv = Gm0;
        else fv = (Gm0 >> sf) | (Gm1 << (64 - sf));
        constexpr int sr = 2 * j;  // reverse value: bits [sr, sr+2K) of [Gc1:Gc0]
        uint64_t rv;
        if constexpr (sr == 0) rv = Gc0;
        else rv = (Gc0 >> sr) | (Gc1 << (64 - sr));
        fv &= MASK2K, rv &= MASK2K;
        uint64_t lt;
        asm("v_cmp_lt_u64_e64 %0, %1, %2" : "=s"(lt) : "v"(rv), "v"(fv));
        asm("v_cndmask_b32_e64 %0, 0, -1, %1" : "=v"(rc_mask) : "s"(lt));
      }
      uint32_t d[ND];
#pragma unroll
      for (int m = 0; m < ND; ++m) {
        uint32_t f;
        if (m < ND - 1) {
          f = (r == 0) ? FA[q + m] : __builtin_amdgcn_alignbyte(FA[q + m + 1], FA[q + m], r);
        } else if (r + NB <= 4) {
          f = (NB == 4) ? FA[q + m] : ((FA[q + m] >> (8 * r)) & ((1u << (8 * (NB & 3))) - 1));
        } else {
          f = __builtin_amdgcn_alignbyte(FA[q + m + 1], FA[q + m], r);
          if (NB < 4) f &= (1u << (8 * (NB & 3))) - 1;
        }
        uint32_t v = f;
        if (CANON) {
          const int e = j + K - 1 - 4 * m;
          const int Q = e >> 2, sft = e & 3;
          uint32_t sel = 0;
#pragma unroll
          for (int i = 0; i < 4; ++i) {
            uint32_t sb = (4 * m + i < K) ? (uint32_t)(sft + 4 - i) : 0x0cu;
            sel |= sb << (8 * i);
          }
          uint32_t rcw = __builtin_amdgcn_perm(CA[Q], (Q >= 1) ? CA[Q - 1] : 0u, sel);
          v = __builtin_amdgcn_bitop3_b32(rc_mask, rcw, f, 0xCA);
        }
        d[m] = v;
      }
      const uint64_t h = t1ha2_fixed<K>(d, seed);
      if (valid && h < threshold) stage_hit(stage, h, gm, g, hits, cnt);
    });
  }
  flush_hits(stage, gm, g, hits, cnt);
}

// 2-bit code of one base (long-k kernel)
__device__ __forceinline__ uint32_t base_code(uint8_t c, uint32_t u2t) {
  // 0..3 for ACGT (either case), 4 otherwise
  uint8_t u = c & 0xDF;
  if (u == 'A') return 0;
  if (u == 'C') return 1;
  if (u == 'G') return 2;
  if (u == 'T') return 3;
  if (u2t && u == 'U') return 3;
  return 4;
}

// =========================================================================================
// long-k kernel: 33 <= k <= 255 (the CPU path's t1ha2 long-input loop; src/cuda_kernel.cu has none)
// =========================================================================================
// Run-time k.  A workgroup stages 1 024 k-mer starts' worth of sequence (<= 1 278 bytes) into LDS three
// ways: the normalised forward strand (upper-case ASCII, 0 for anything that is not a base), its reverse
// complement (so that the reverse strand of a k-mer is an ascending byte range too), and a running count of
// invalid bytes (a window is valid iff the count does not change across it).  Each lane then takes four
// starts: strand choice by the first differing dword of the two byte strings (big-endian compare = the
// reference's lexicographic compare), hash words cut out of LDS with a run-time v_alignbyte funnel.
constexpr uint32_t LONG_TILE = 1024;                 // k-mer starts per tile
constexpr uint32_t LONG_BYTES = 1536;                // staged bytes per tile (>= LONG_TILE + 254), 6 per lane
constexpr uint32_t LONG_DW = LONG_BYTES / 4 + 4;     // + zero slack for the funnel's second dword

struct LdsStrand {
  const uint32_t *w32;  // LDS array of bytes, dword view
  uint32_t byte0;       // first byte of the k-mer in that array
  __device__ __forceinline__ uint32_t dword(uint32_t i) const {  // bytes [byte0 + 4i, byte0 + 4i + 4)
    const uint32_t o = byte0 + 4 * i, a = o >> 2;
    return __builtin_amdgcn_alignbyte(w32[a + 1], w32[a], o & 3u);
  }
  __device__ __forceinline__ uint64_t word(uint32_t byte_off, uint32_t nbytes) const {  // little endian, byte_off % 8 == 0
    uint32_t lo = dword(byte_off / 4), hi = nbytes > 4 ? dword(byte_off / 4 + 1) : 0u;
    if (nbytes < 4) lo &= (1u << (8 * nbytes)) - 1;
    else if (nbytes > 4 && nbytes < 8) hi &= (1u << (8 * (nbytes - 4))) - 1;
    return mk64(lo, hi);
  }
};

__device__ __forceinline__ uint64_t t1ha2_long(const LdsStrand &sb, uint32_t len, uint64_t seed) {
  uint64_t a = seed, b = (uint64_t)len;
  uint32_t off = 0;
  if (len > 32) {  // published t1ha2: lanes c,d, 32 bytes per round, squash
    uint64_t c = rot64((uint64_t)len, 23) + ~seed;
    uint64_t d = ~(uint64_t)len + rot64(seed, 19);
    do {
      const uint64_t w0 = sb.word(off, 8), w1 = sb.word(off + 8, 8), w2 = sb.word(off + 16, 8), w3 = sb.word(off + 24, 8);
      off += 32;
      const uint64_t d02 = w0 + rot64(w2 + d, 56);
      const uint64_t c13 = w1 + rot64(w3 + c, 19);
      d ^= b + rot64(w1, 38);
      c ^= a + rot64(w0, 57);
      b ^= P6 * (c13 + w2);
      a ^= P5 * (d02 + w3);
    } while (off + 31 < len);  // data < detent  <=>  off < len - 31
    a ^= P6 * (c + rot64(d, 23));
    b ^= P5 * (rot64(c, 19) + d);
    len &= 31;
  }
  uint32_t rem = len;
  if (rem > 24) { mixup64<P4>(a, b, sb.word(off, 8)); off += 8, rem -= 8; }
  if (rem > 16) { mixup64<P3>(b, a, sb.word(off, 8)); off += 8, rem -= 8; }
  if (rem > 8) { mixup64<P2>(a, b, sb.word(off, 8)); off += 8, rem -= 8; }
  if (rem > 0) mixup64<P1>(b, a, sb.word(off, rem));
  return final64(a, b);
}

__global__ __launch_bounds__(WG) void kmer_sample_long(
    const uint8_t *__restrict__ seq, const hg_genome_meta *__restrict__ meta,
    const uint32_t *__restrict__ item_genome, uint32_t ksize, uint64_t threshold, uint64_t seed,
    uint32_t canonical, uint32_t u2t, uint64_t *__restrict__ hits, uint32_t *__restrict__ cnt) {
  __shared__ HitStage stage;
  __shared__ uint32_t s_f[LONG_DW], s_rc[LONG_DW];
  __shared__ uint16_t s_bad[LONG_BYTES + 8];  // s_bad[i] = invalid bytes among the first i staged bytes
  __shared__ uint16_t s_scan[WG];
  const uint32_t item = blockIdx.x, tid = threadIdx.x;
  const uint32_t g = item_genome[item];
  const hg_genome_meta gm = meta[g];
  const uint64_t n_bps = gm.n_bps;
  if (n_bps < ksize) return;
  const uint64_t n_starts = n_bps - ksize + 1;
  const uint8_t *__restrict__ gseq = seq + gm.seq_off;
  const uint64_t item_start = (uint64_t)(item - gm.item_first) * GEN_ITEM;
  if (tid == 0) stage.n = 0;
  const uint32_t n_stage = LONG_TILE + ksize - 1;  // bytes a full tile needs
  uint8_t *fb = reinterpret_cast<uint8_t *>(s_f), *rb = reinterpret_cast<uint8_t *>(s_rc);

  for (uint64_t tile0 = item_start; tile0 < item_start + GEN_ITEM && tile0 < n_starts; tile0 += LONG_TILE) {
    __syncthreads();  // previous tile's readers are done
    for (uint32_t i = tid; i < LONG_DW; i += WG) s_f[i] = 0u, s_rc[i] = 0u;
    __syncthreads();
    // ---- stage: 6 consecutive bytes per lane
    uint32_t nbad = 0;
    uint8_t fbyte[6];
#pragma unroll
    for (int j = 0; j < 6; ++j) {
      const uint32_t i = tid * 6 + j;
      const uint64_t pos = tile0 + i;
      uint32_t code = 4;
      if (i < n_stage && pos < n_bps) code = base_code(gseq[pos], u2t);
      fbyte[j] = code < 4 ? (uint8_t)(0x54474341u >> (8 * code)) : (uint8_t)0;  // "ACGT"
      nbad += code < 4 ? 0u : 1u;
      if (i < n_stage) {
        fb[i] = fbyte[j];
        rb[n_stage - 1 - i] = code < 4 ? (uint8_t)(0x41434754u >> (8 * code)) : (uint8_t)0;  // "TGCA"
      }
    }
    // exclusive scan of the per-lane invalid counts (Hillis-Steele over WG entries)
    s_scan[tid] = (uint16_t)nbad;
    __syncthreads();
    for (uint32_t o = 1; o < WG; o <<= 1) {
      const uint16_t add = tid >= o ? s_scan[tid - o] : (uint16_t)0;
      __syncthreads();
      s_scan[tid] = (uint16_t)(s_scan[tid] + add);
      __syncthreads();
    }
    {
      uint32_t run = (uint32_t)s_scan[tid] - nbad;  // invalid bytes before this lane's first byte
#pragma unroll
      for (int j = 0; j < 6; ++j) {
        s_bad[tid * 6 + j] = (uint16_t)run;
        run += fbyte[j] ? 0u : 1u;
      }
      if (tid == WG - 1) s_bad[LONG_BYTES] = (uint16_t)run;
    }
    __syncthreads();
    // ---- the lane's four starts
#pragma unroll 1
    for (uint32_t j = 0; j < LONG_TILE / WG; ++j) {
      const uint32_t p = tid + WG * j;
      if (tile0 + p >= n_starts) break;
      if (s_bad[p + ksize] != s_bad[p]) continue;  // a non-base inside the window
      const LdsStrand f{s_f, p}, r{s_rc, n_stage - p - ksize};
      bool use_rc = false;
      if (canonical) {  // first differing dword decides (big-endian = lexicographic on the bytes)
        for (uint32_t t = 0; 4 * t < ksize; ++t) {
          uint32_t fw = __builtin_bswap32(f.dword(t)), rw = __builtin_bswap32(r.dword(t));
          const uint32_t left = ksize - 4 * t;
          if (left < 4) fw &= ~0u << (8 * (4 - left)), rw &= ~0u << (8 * (4 - left));
          if (fw != rw) {
            use_rc = rw < fw;
            break;
          }
        }
      }
      const uint64_t h = t1ha2_long(use_rc ? r : f, ksize, seed);
      if (h < threshold) stage_hit(stage, h, gm, g, hits, cnt);
    }
  }
  flush_hits(stage, gm, g, hits, cnt);
}

template <int K>
hipError_t launch_fast(hipStream_t st, bool canonical, uint32_t n_items, const uint8_t *d_seq,
                       const hg_genome_meta *d_meta, const uint32_t *d_item_genome, uint64_t threshold,
                       uint64_t seed, uint32_t u2t, uint64_t *d_hits, uint32_t *d_cnt) {
#ifdef HG_KMER_EXPERIMENT
  if (K == 21 && canonical) {  // development switch: HG_KMER_VARIANT=0..3
    const char *e = getenv("HG_KMER_VARIANT");
    const int v = e ? atoi(e) : 0;
#define HG_V(VV)                                                                                         \
  case VV:                                                                                               \
    hipLaunchKernelGGL((kmer_sample_fast<21, true, VV>), dim3(n_items), dim3(WG), 0, st, d_seq, d_meta, \
                       d_item_genome, threshold, seed, u2t, d_hits, d_cnt);                              \
    return hipGetLastError();
    switch (v) { HG_V(0) HG_V(1) HG_V(2) HG_V(3) HG_V(4) HG_V(12) HG_V(28) HG_V(60) default: break; }
#undef HG_V
  }
#endif
  if (HG_KMER_SHARED) {
    if (canonical)
      hipLaunchKernelGGL((kmer_sample_shared<K, true>), dim3(n_items), dim3(WG), 0, st, d_seq, d_meta, d_item_genome, threshold,
                         seed, u2t, d_hits, d_cnt);
    else
      hipLaunchKernelGGL((kmer_sample_shared<K, false>), dim3(n_items), dim3(WG), 0, st, d_seq, d_meta, d_item_genome, threshold,
                         seed, u2t, d_hits, d_cnt);
    return hipGetLastError();
  }
  if constexpr (Geo<K>::M == 12) {  // k = 18..21: three slices of 12 k-mers per window
    if (canonical && HG_KMER_GROUPED) {
      hipLaunchKernelGGL((kmer_sample_grouped<K>), dim3(n_items), dim3(WG), 0, st, d_seq, d_meta, d_item_genome, threshold,
                         seed, u2t, d_hits, d_cnt);
      return hipGetLastError();
    }
  }
  if (canonical)
    hipLaunchKernelGGL((kmer_sample_fast<K, true, HG_KMER_DEFAULT_VAR>), dim3(n_items), dim3(WG), 0, st, d_seq, d_meta,
                       d_item_genome, threshold, seed, u2t, d_hits, d_cnt);
  else
    hipLaunchKernelGGL((kmer_sample_fast<K, false>), dim3(n_items), dim3(WG), 0, st, d_seq, d_meta,
                       d_item_genome, threshold, seed, u2t, d_hits, d_cnt);
  return hipGetLastError();
}

}  // namespace

const char *hg_kmer_kernel_name(uint32_t k, bool canonical) {  // mirrors hg_launch_kmer_sample / launch_fast
  static thread_local char buf[64];
  if (k > 32) return "kmer_sample_long";
  if (HG_KMER_SHARED && k < FAST64_FROM) {
    snprintf(buf, sizeof buf, "kmer_sample_shared<%u, %s>", k, canonical ? "true" : "false");
    return buf;
  }
  const bool grouped = canonical && HG_KMER_GROUPED && k >= 18 && k <= 25;
  if (grouped) snprintf(buf, sizeof buf, "kmer_sample_grouped<%u>", k);
  else if (k >= 22) snprintf(buf, sizeof buf, "kmer_sample_fast64<%u, %s>", k, canonical ? "true" : "false");
  else if (canonical) snprintf(buf, sizeof buf, "kmer_sample_fast<%u, true, %d>", k, (int)HG_KMER_DEFAULT_VAR);
  else snprintf(buf, sizeof buf, "kmer_sample_fast<%u, false, 0>", k);
  return buf;
}

uint32_t hg_kmer_item_starts(uint32_t k) {
  if (fast64_k(k)) return (uint32_t)(WG * 32 * TILES_PER_ITEM64);
  if (!fast_k(k)) return GEN_ITEM;
  if (HG_KMER_SHARED) return (uint32_t)GeoS<21>::ITEM;  // (the same for every k <= 21)
  return (uint32_t)(WG * ((33 - k) & ~3u) * tiles_per_item((int)k));
}

hipError_t hg_launch_kmer_sample(hipStream_t st, const uint8_t *d_seq, const hg_genome_meta *d_meta,
                                 const uint32_t *d_item_genome, uint32_t n_items, uint32_t ksize,
                                 uint64_t threshold, uint64_t seed, bool canonical, uint32_t norm_mode,
                                 uint64_t *d_hits, uint32_t *d_cnt) {
  if (n_items == 0) return hipSuccess;
  const uint32_t u2t = (norm_mode == HG_NORM_U2T) ? 1u : 0u;
#define HG_FAST_CASE(KK)                                                                       \
  case KK:                                                                                     \
    return launch_fast<KK>(st, canonical, n_items, d_seq, d_meta, d_item_genome, threshold,   \
                           seed, u2t, d_hits, d_cnt);
  if (fast_k(ksize)) switch (ksize) {
    HG_FAST_CASE(1) HG_FAST_CASE(2) HG_FAST_CASE(3) HG_FAST_CASE(4) HG_FAST_CASE(5) HG_FAST_CASE(6) HG_FAST_CASE(7)
    HG_FAST_CASE(8) HG_FAST_CASE(9) HG_FAST_CASE(10) HG_FAST_CASE(11) HG_FAST_CASE(12) HG_FAST_CASE(13)
    HG_FAST_CASE(14) HG_FAST_CASE(15) HG_FAST_CASE(16) HG_FAST_CASE(17) HG_FAST_CASE(18)
    HG_FAST_CASE(19) HG_FAST_CASE(20) HG_FAST_CASE(21)
    default:
      break;
  }
#undef HG_FAST_CASE
#define HG_FAST64_CASE(KK)                                                                              \
  case KK:                                                                                              \
    if constexpr (Geo<KK>::M == 8) { /* k = 22..25: four slices of 8 k-mers per 56-base window */      \
      if (canonical && HG_KMER_GROUPED) {                                                               \
        hipLaunchKernelGGL((kmer_sample_grouped<KK>), dim3(n_items), dim3(WG), 0, st, d_seq, d_meta,    \
                           d_item_genome, threshold, seed, u2t, d_hits, d_cnt);                         \
        return hipGetLastError();                                                                       \
      }                                                                                                 \
    }                                                                                                   \
    if (canonical)                                                                                      \
      hipLaunchKernelGGL((kmer_sample_fast64<KK, true>), dim3(n_items), dim3(WG), 0, st, d_seq, d_meta, \
                         d_item_genome, threshold, seed, u2t, d_hits, d_cnt);                           \
    else                                                                                                \
      hipLaunchKernelGGL((kmer_sample_fast64<KK, false>), dim3(n_items), dim3(WG), 0, st, d_seq, d_meta, \
                         d_item_genome, threshold, seed, u2t, d_hits, d_cnt);                           \
    return hipGetLastError();
  switch (ksize) {
    HG_FAST64_CASE(22) HG_FAST64_CASE(23) HG_FAST64_CASE(24) HG_FAST64_CASE(25) HG_FAST64_CASE(26)
    HG_FAST64_CASE(27) HG_FAST64_CASE(28) HG_FAST64_CASE(29)
    HG_FAST64_CASE(30) HG_FAST64_CASE(31) HG_FAST64_CASE(32)
    default:
      break;
  }
#undef HG_FAST64_CASE
  // 33 <= k <= 255 (k <= 32 returned above)
  hipLaunchKernelGGL(kmer_sample_long, dim3(n_items), dim3(WG), 0, st, d_seq, d_meta, d_item_genome, ksize,
                     threshold, seed, canonical ? 1u : 0u, u2t, d_hits, d_cnt);
  return hipGetLastError();
}
