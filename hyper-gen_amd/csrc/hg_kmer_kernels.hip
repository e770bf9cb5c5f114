// hg_kmer_kernels.hip -- FracMinHash k-mer hash + threshold sample on gfx950.
//
// Computes what extract_kmer_hash (src/sketch.rs:71-98) / cuda_kmer_t1ha2
// (src/cuda_kernel.cu:250-321) compute: for every k-window of ACGT bases the canonical
// strand's ASCII bytes are hashed with t1ha2_atonce(seed) and hashes below the threshold
// are kept.  The design is MI355X-first, not the reference's (1 thread x 512 k-mers with
// two 544-byte scratch arrays):
//
//  * kmer_sample_shared<K, CANON> (k = 1..32): a workgroup stages its tile -- 254 lanes x 12 k-mer starts -- in LDS
//    once, classified 4 bases per instruction (SWAR on dwords: 2-bit codes, upper-cased ASCII and complement ASCII
//    from v_perm_b32 lookups, validity from one XOR), as a forward image and a reverse-complement image in all four
//    byte phases, so that both strands of every k-mer are runs of whole dwords;
//  * the canonical strand is chosen by ONE 64-bit compare of 2-bit packed k-mers (A<C<G<T holds both in ASCII and in
//    the 2-bit code, so this equals the reference's byte-wise compare, src/cuda_kernel.cu:306-311) and becomes an
//    ADDRESS: one v_cndmask picks the image, the hash words are ds_reads with immediate offsets -- no instruction
//    touches the bytes;
//  * t1ha2 is specialised per k; for k = 17..32 the whole k-mer body (word reads, three or four mixup64 stages,
//    final64, threshold compare) is one asm statement on fixed register pairs: 57 / 67 vector instructions, 25 / 30 of
//    them v_mad_u64_u32;
//  * survivors (1/scaled of the k-mers) are staged in a small LDS list and the workgroup reserves its range of the
//    genome's hit slice with ONE global atomic at the end of its work item; lossless: no 8-slot cap like
//    src/cuda_kernel.cu:316, hash value 0 is kept;
//  * kmer_sample_long (k = 33..255): run-time k, t1ha2's long-input loop.
// Predecessors, removed from the source after the shared-image kernel replaced them for every k (DESIGN.md 4.1 has
// their measurements): kmer_sample_fast (32-base register windows), kmer_sample_grouped (56-base windows, per-lane
// LDS images, run-time byte shifts), kmer_sample_fast64 (64-base windows).
//
// The kernel is integer-VALU bound, not HBM bound.
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

// t1ha2_atonce for a compile-time length K <= 32 on 8-byte little-endian words w[0..ceil(K/8)) with the unused bytes
// of the last word zero (tail switch of src/cuda_kernel.cu:205-245)
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
constexpr int WG = 256;                              // lanes per workgroup
constexpr int GEN_STARTS = 48;                       // long-k kernel: starts per lane and work item (eight tiles of 1 536)
constexpr int GEN_ITEM = WG * GEN_STARTS;            // and per work item

// One hit straight to the genome's list.  The lanes of a wave that arrive here together (a divergent branch: g is
// wave-uniform) reserve their slots with ONE atomic: the counters of a genome are one address for all of its work items,
// and per-lane atomics on it serialise at ~3.5 ns each (1 000 x 5 Mbp at scaled = 50, when every work item overflowed its
// LDS list by 300 hits: 194 ms instead of 9).
__device__ __forceinline__ void append_hit(uint64_t h, const hg_genome_meta &gm, uint32_t g,
                                           uint64_t *__restrict__ hits, uint32_t *__restrict__ cnt) {
  const unsigned long long m = __ballot(1);  // the lanes in this branch
  const uint32_t lane = threadIdx.x & 63u;
  uint32_t base = 0;
  if (lane == (uint32_t)__builtin_ctzll(m)) base = atomicAdd(&cnt[g], (uint32_t)__popcll(m));
  base = __builtin_amdgcn_readfirstlane(base);  // (the first active lane is the one that asked)
  const uint32_t idx = base + (uint32_t)__popcll(m & ((1ull << lane) - 1ull));
  if (idx < gm.hit_cap) hits[gm.hit_off + idx] = h;
}

// Workgroup-level staging of the sampled hashes (fast kernels).  A hit is 1 k-mer in `scaled`, but every
// one used to cost its wave a returning global atomic -- a full memory round trip in the middle of ~1 400
// VALU instructions, about once per two tiles per wave.  Hits go to an LDS list instead (LDS atomic) and the
// workgroup reserves its range of the genome's hit buffer once, at the end of its work item.  A list that
// overflows (low-complexity sequence: every position of a repeat samples the same hash) spills straight to
// the global path; the raw counter semantics (it keeps counting past the capacity) are unchanged.
constexpr uint32_t HIT_STAGE = 256, HIT_STAGE_MAX = 4096;
struct HitStage {
  uint32_t n, base;
};
// The list itself is the kernels' dynamic LDS: `cap` entries, HIT_STAGE at the default sampling rate and up to HIT_STAGE_MAX
// when a tile alone yields hundreds of hits (hg_launch_kmer_sample sizes it: scaled = 5 samples 610 of a tile's 3 048
// starts, and a list of 256 sent the rest through a global atomic per wave and k-mer step -- 168 ms for 1 000 x 5 Mbp).
extern __shared__ __attribute__((aligned(8))) uint64_t s_stage_h[];
__device__ __forceinline__ void stage_hit(HitStage &st, uint32_t cap, uint64_t h, const hg_genome_meta &gm, uint32_t g,
                                          uint64_t *__restrict__ hits, uint32_t *__restrict__ cnt) {
  const uint32_t idx = atomicAdd(&st.n, 1u);
  if (idx < cap) s_stage_h[idx] = h;
  else append_hit(h, gm, g, hits, cnt);
}
// AGAIN: the list is emptied between two tiles and filled again afterwards (the caller's next barrier lies between this
// call's reads and the next writes)
template <bool AGAIN = false>
__device__ __forceinline__ void flush_hits(HitStage &st, uint32_t cap, const hg_genome_meta &gm, uint32_t g,
                                           uint64_t *__restrict__ hits, uint32_t *__restrict__ cnt) {
  __syncthreads();
  const uint32_t n = st.n < cap ? st.n : cap;
  if (threadIdx.x == 0 && n) st.base = atomicAdd(&cnt[g], n);
  __syncthreads();
  if (AGAIN && threadIdx.x == 0) st.n = 0;  // (every lane has read it)
  for (uint32_t i = threadIdx.x; i < n; i += blockDim.x) {
    const uint32_t idx = st.base + i;
    if (idx < gm.hit_cap) hits[gm.hit_off + idx] = s_stage_h[i];
  }
}
// Between two tiles (behind the barrier that ends a tile: st.n is final and the same for every lane): a list that is a
// quarter full goes out now.  At the default sampling rate a work item collects ~18 hits and never gets here; a denser
// sketch (scaled = 100: 270 hits per item) used to overflow the list and pay a global atomic per hit.
__device__ __forceinline__ void flush_hits_if_filling(HitStage &st, uint32_t cap, const hg_genome_meta &gm, uint32_t g,
                                                      uint64_t *__restrict__ hits, uint32_t *__restrict__ cnt) {
  if (st.n >= cap / 4) flush_hits<true>(st, cap, gm, g, hits, cnt);  // workgroup-uniform
}
// Whether the sampling rate can fill a quarter of the list within one work item at all (a kernel argument: the test
// between the tiles is a scalar branch that the default rate, 1 in 1 500, never takes -- the LDS read of the fill level
// cost the headline 0.2 %): more than ~32 expected hits per 27 000 starts, i.e. scaled < 850.  (A repeat that piles more hits than that into an
// item of a sparse sketch overflows the list as before; those hits go out one atomic per wave.)
__device__ __forceinline__ bool dense_sampling(uint64_t threshold) { return threshold > (~0ull / 850ull); }

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
// request the words of one k-mer into one buffer (registers R0..R7 = four pairs; three of them for K <= 24): whole
// dwords of the strand's phase image and the last 1..4 bytes.  Cases C = 10 * dwords + bytes in the last dword:
// K = 17..20: 51..54, 21..24: 61..64, 25..28: 71..74, 29..32: 81..84.
// (dword reads: a ds_read_b64 at an address that is a multiple of 4 but not of 8 is as slow as a byte-aligned one --
// measured 18.8 ms against 8.5 for the whole kernel -- and ds_read2_b32's 8-bit offsets cannot hold the phase image's)
#define HG_KS_RD(R, O) "ds_read_b32 v" R ", %[base] offset:%[" O "]\n\t"
#define HG_KS_RD4(R, O) "ds_read_b32 v" R ", %[base] offset:%[" O "] + 4\n\t"
#define HG_KS_HEAD2(R0, R1, R2, R3) HG_KS_RD(R0, "o0") HG_KS_RD4(R1, "o0") HG_KS_RD(R2, "o1") HG_KS_RD4(R3, "o1")
#define HG_KS_HEAD3(R0, R1, R2, R3, R4, R5) HG_KS_HEAD2(R0, R1, R2, R3) HG_KS_RD(R4, "o2") HG_KS_RD(R5, "o3")
// the last pair: odd dword count (low half only, high half zero) / even (a whole dword + the rest)
#define HG_KS_TAIL_O1(RL, RH, OL, OH) "ds_read_u8 v" RL ", %[base] offset:%[" OL "]\n\tv_mov_b32 v" RH ", 0"
#define HG_KS_TAIL_O2(RL, RH, OL, OH) "ds_read_u16 v" RL ", %[base] offset:%[" OL "]\n\tv_mov_b32 v" RH ", 0"
#define HG_KS_TAIL_O3(RL, RH, OL, OH) "ds_read_b32 v" RL ", %[base] offset:%[" OL "]\n\tv_mov_b32 v" RH ", 0"
#define HG_KS_TAIL_O4(RL, RH, OL, OH) HG_KS_TAIL_O3(RL, RH, OL, OH)
#define HG_KS_TAIL_E1(RL, RH, OL, OH) HG_KS_RD(RL, OL) "ds_read_u8 v" RH ", %[base] offset:%[" OH "]"
#define HG_KS_TAIL_E2(RL, RH, OL, OH) HG_KS_RD(RL, OL) "ds_read_u16 v" RH ", %[base] offset:%[" OH "]"
#define HG_KS_TAIL_E3(RL, RH, OL, OH) HG_KS_RD(RL, OL) "ds_read_b32 v" RH ", %[base] offset:%[" OH "]"
#define HG_KS_TAIL_E4(RL, RH, OL, OH) HG_KS_TAIL_E3(RL, RH, OL, OH)
#define HG_KS_READ_TEXT_51(R0, R1, R2, R3, R4, R5, R6, R7) HG_KS_HEAD2(R0, R1, R2, R3) HG_KS_TAIL_O1(R4, R5, "o2", "o3")
#define HG_KS_READ_TEXT_52(R0, R1, R2, R3, R4, R5, R6, R7) HG_KS_HEAD2(R0, R1, R2, R3) HG_KS_TAIL_O2(R4, R5, "o2", "o3")
#define HG_KS_READ_TEXT_53(R0, R1, R2, R3, R4, R5, R6, R7) HG_KS_HEAD2(R0, R1, R2, R3) HG_KS_TAIL_O3(R4, R5, "o2", "o3")
#define HG_KS_READ_TEXT_54(R0, R1, R2, R3, R4, R5, R6, R7) HG_KS_HEAD2(R0, R1, R2, R3) HG_KS_TAIL_O4(R4, R5, "o2", "o3")
#define HG_KS_READ_TEXT_61(R0, R1, R2, R3, R4, R5, R6, R7) HG_KS_HEAD2(R0, R1, R2, R3) HG_KS_TAIL_E1(R4, R5, "o2", "o3")
#define HG_KS_READ_TEXT_62(R0, R1, R2, R3, R4, R5, R6, R7) HG_KS_HEAD2(R0, R1, R2, R3) HG_KS_TAIL_E2(R4, R5, "o2", "o3")
#define HG_KS_READ_TEXT_63(R0, R1, R2, R3, R4, R5, R6, R7) HG_KS_HEAD2(R0, R1, R2, R3) HG_KS_TAIL_E3(R4, R5, "o2", "o3")
#define HG_KS_READ_TEXT_64(R0, R1, R2, R3, R4, R5, R6, R7) HG_KS_HEAD2(R0, R1, R2, R3) HG_KS_TAIL_E4(R4, R5, "o2", "o3")
#define HG_KS_READ_TEXT_71(R0, R1, R2, R3, R4, R5, R6, R7) HG_KS_HEAD3(R0, R1, R2, R3, R4, R5) HG_KS_TAIL_O1(R6, R7, "o4", "o5")
#define HG_KS_READ_TEXT_72(R0, R1, R2, R3, R4, R5, R6, R7) HG_KS_HEAD3(R0, R1, R2, R3, R4, R5) HG_KS_TAIL_O2(R6, R7, "o4", "o5")
#define HG_KS_READ_TEXT_73(R0, R1, R2, R3, R4, R5, R6, R7) HG_KS_HEAD3(R0, R1, R2, R3, R4, R5) HG_KS_TAIL_O3(R6, R7, "o4", "o5")
#define HG_KS_READ_TEXT_74(R0, R1, R2, R3, R4, R5, R6, R7) HG_KS_HEAD3(R0, R1, R2, R3, R4, R5) HG_KS_TAIL_O4(R6, R7, "o4", "o5")
#define HG_KS_READ_TEXT_81(R0, R1, R2, R3, R4, R5, R6, R7) HG_KS_HEAD3(R0, R1, R2, R3, R4, R5) HG_KS_TAIL_E1(R6, R7, "o4", "o5")
#define HG_KS_READ_TEXT_82(R0, R1, R2, R3, R4, R5, R6, R7) HG_KS_HEAD3(R0, R1, R2, R3, R4, R5) HG_KS_TAIL_E2(R6, R7, "o4", "o5")
#define HG_KS_READ_TEXT_83(R0, R1, R2, R3, R4, R5, R6, R7) HG_KS_HEAD3(R0, R1, R2, R3, R4, R5) HG_KS_TAIL_E3(R6, R7, "o4", "o5")
#define HG_KS_READ_TEXT_84(R0, R1, R2, R3, R4, R5, R6, R7) HG_KS_HEAD3(R0, R1, R2, R3, R4, R5) HG_KS_TAIL_E4(R6, R7, "o4", "o5")
// wait for the words requested one hash ago; K % 4 == 3: the last dword carries a byte of the next base
#define HG_KS_W "s_waitcnt lgkmcnt(0)\n\t"
#define HG_KS_WM(R) HG_KS_W "v_and_b32 v" R ", 0xffffff, v" R "\n\t"
#define HG_KS_WAIT_TEXT_51(R0, R1, R2, R3, R4, R5, R6, R7) HG_KS_W
#define HG_KS_WAIT_TEXT_52(R0, R1, R2, R3, R4, R5, R6, R7) HG_KS_W
#define HG_KS_WAIT_TEXT_53(R0, R1, R2, R3, R4, R5, R6, R7) HG_KS_WM(R4)
#define HG_KS_WAIT_TEXT_54(R0, R1, R2, R3, R4, R5, R6, R7) HG_KS_W
#define HG_KS_WAIT_TEXT_61(R0, R1, R2, R3, R4, R5, R6, R7) HG_KS_W
#define HG_KS_WAIT_TEXT_62(R0, R1, R2, R3, R4, R5, R6, R7) HG_KS_W
#define HG_KS_WAIT_TEXT_63(R0, R1, R2, R3, R4, R5, R6, R7) HG_KS_WM(R5)
#define HG_KS_WAIT_TEXT_64(R0, R1, R2, R3, R4, R5, R6, R7) HG_KS_W
#define HG_KS_WAIT_TEXT_71(R0, R1, R2, R3, R4, R5, R6, R7) HG_KS_W
#define HG_KS_WAIT_TEXT_72(R0, R1, R2, R3, R4, R5, R6, R7) HG_KS_W
#define HG_KS_WAIT_TEXT_73(R0, R1, R2, R3, R4, R5, R6, R7) HG_KS_WM(R6)
#define HG_KS_WAIT_TEXT_74(R0, R1, R2, R3, R4, R5, R6, R7) HG_KS_W
#define HG_KS_WAIT_TEXT_81(R0, R1, R2, R3, R4, R5, R6, R7) HG_KS_W
#define HG_KS_WAIT_TEXT_82(R0, R1, R2, R3, R4, R5, R6, R7) HG_KS_W
#define HG_KS_WAIT_TEXT_83(R0, R1, R2, R3, R4, R5, R6, R7) HG_KS_WM(R7)
#define HG_KS_WAIT_TEXT_84(R0, R1, R2, R3, R4, R5, R6, R7) HG_KS_W
// the two word buffers: (R0..R7) and the same as register pairs; K <= 24 uses the first three pairs
#define HG_KS_REGS0 ("56", "57", "58", "59", "60", "61", "90", "91")
#define HG_KS_REGS1 ("62", "63", "64", "65", "66", "67", "92", "93")
#define HG_KS_PAIRS0 ("v[56:57]", "v[58:59]", "v[60:61]", "v[90:91]")
#define HG_KS_PAIRS1 ("v[62:63]", "v[64:65]", "v[66:67]", "v[92:93]")
#define HG_KS_APPLY(M, ARGS) M ARGS
// final64(a, b): x = (a + rot64(b, 41)) * P0, y = (rot64(a, 23) + b) * P6 (low halves), z = x ^ y, mux64(z, P5), compare
#define HG_KS_FINAL(AP, AL, AH, BP, BL, BH)                                                                          \
  "v_alignbit_b32 v86, " BL ", " BH ", 9\n\t"                                                                        \
  "v_alignbit_b32 v87, " BH ", " BL ", 9\n\t"                                                                        \
  "v_lshl_add_u64 v[72:73], v[86:87], 0, " AP "\n\t"                                                                 \
  "v_alignbit_b32 v86, " AH ", " AL ", 23\n\t"                                                                       \
  "v_alignbit_b32 v87, " AL ", " AH ", 23\n\t"                                                                       \
  "v_lshl_add_u64 v[88:89], v[86:87], 0, " BP "\n\t"                                                                 \
  HG_KS_LO64("v72", "v73", "%[p0l]", "%[p0h]", "v[74:75]", "v74", "v75")                                             \
  HG_KS_LO64("v88", "v89", "%[p6l]", "%[p6h]", "v[84:85]", "v84", "v85")                                             \
  "v_xor_b32_e32 v72, v74, v84\n\t"                                                                                  \
  "v_xor_b32_e32 v73, v75, v85\n\t"                                                                                  \
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
// 17 <= K <= 24: three stages
#define HG_KS_HASH_TEXT_3(W0, W1, W2, W3)                                                                            \
  /* mixup64<P3>(b, a, w0): x = seed + w0; b = K ^ lo; a = seed + hi */                                              \
  "v_lshl_add_u64 v[72:73], " W0 ", 0, %[seed]\n\t"                                                                  \
  HG_KS_MUL128("v72", "v73", "%[p3l]", "%[p3h]", "%[seed]", "v[78:79]", "v79")                                       \
  "v_xor_b32_e32 v80, %[kk], v74\n\t"                                                                                \
  "v_mov_b32 v81, v76\n\t"                                                                                           \
  /* mixup64<P2>(a, b, w1): x = b + w1; a ^= lo; b += hi   (a = U, b = S1 -> a = S2, b = R) */                       \
  "v_lshl_add_u64 v[72:73], v[80:81], 0, " W1 "\n\t"                                                                 \
  HG_KS_MUL128("v72", "v73", "%[p2l]", "%[p2h]", "v[80:81]", "v[84:85]", "v85")                                      \
  "v_xor_b32_e32 v82, v78, v74\n\t"                                                                                  \
  "v_xor_b32_e32 v83, v79, v76\n\t"                                                                                  \
  /* mixup64<P1>(b, a, w2): x = a + w2; b ^= lo; a += hi   (a = S2, b = R -> b = S1, a = U) */                       \
  "v_lshl_add_u64 v[72:73], v[82:83], 0, " W2 "\n\t"                                                                 \
  HG_KS_MUL128("v72", "v73", "%[p1l]", "%[p1h]", "v[82:83]", "v[78:79]", "v79")                                      \
  "v_xor_b32_e32 v80, v84, v74\n\t"                                                                                  \
  "v_xor_b32_e32 v81, v85, v76\n\t"                                                                                  \
  HG_KS_FINAL("v[78:79]", "v78", "v79", "v[80:81]", "v80", "v81")
// 25 <= K <= 32: four stages (the first one's b is the length, a small constant)
#define HG_KS_HASH_TEXT_4(W0, W1, W2, W3)                                                                            \
  /* mixup64<P4>(a, b, w0): x = K + w0; a = seed ^ lo; b = K + hi   (-> a = S1, b = U) */                            \
  "v_lshl_add_u64 v[72:73], " W0 ", 0, %[kk]\n\t"                                                                    \
  HG_KS_MUL128("v72", "v73", "%[p4l]", "%[p4h]", "%[kk]", "v[78:79]", "v79")                                         \
  "v_xor_b32_e32 v80, %[seedl], v74\n\t"                                                                             \
  "v_xor_b32_e32 v81, %[seedh], v76\n\t"                                                                             \
  /* mixup64<P3>(b, a, w1): x = a + w1; b ^= lo; a += hi   (a = S1, b = U -> b = S2, a = R) */                       \
  "v_lshl_add_u64 v[72:73], v[80:81], 0, " W1 "\n\t"                                                                 \
  HG_KS_MUL128("v72", "v73", "%[p3l]", "%[p3h]", "v[80:81]", "v[84:85]", "v85")                                      \
  "v_xor_b32_e32 v82, v78, v74\n\t"                                                                                  \
  "v_xor_b32_e32 v83, v79, v76\n\t"                                                                                  \
  /* mixup64<P2>(a, b, w2): x = b + w2; a ^= lo; b += hi   (a = R, b = S2 -> a = S1, b = U) */                       \
  "v_lshl_add_u64 v[72:73], v[82:83], 0, " W2 "\n\t"                                                                 \
  HG_KS_MUL128("v72", "v73", "%[p2l]", "%[p2h]", "v[82:83]", "v[78:79]", "v79")                                      \
  "v_xor_b32_e32 v80, v84, v74\n\t"                                                                                  \
  "v_xor_b32_e32 v81, v85, v76\n\t"                                                                                  \
  /* mixup64<P1>(b, a, w3): x = a + w3; b ^= lo; a += hi   (a = S1, b = U -> b = S2, a = R) */                       \
  "v_lshl_add_u64 v[72:73], v[80:81], 0, " W3 "\n\t"                                                                 \
  HG_KS_MUL128("v72", "v73", "%[p1l]", "%[p1h]", "v[80:81]", "v[84:85]", "v85")                                      \
  "v_xor_b32_e32 v82, v78, v74\n\t"                                                                                  \
  "v_xor_b32_e32 v83, v79, v76\n\t"                                                                                  \
  HG_KS_FINAL("v[84:85]", "v84", "v85", "v[82:83]", "v82", "v83")
#define HG_KS_HASH_INPUTS                                                                                            \
  [seed] "s"(seed), [thr] "s"(threshold), [kk] "n"(K), [p0l] "s"((uint32_t)P0), [p0h] "s"((uint32_t)(P0 >> 32)),     \
      [p1l] "s"((uint32_t)P1), [p1h] "s"((uint32_t)(P1 >> 32)), [p2l] "s"((uint32_t)P2), [p2h] "s"((uint32_t)(P2 >> 32)), \
      [p3l] "s"((uint32_t)P3), [p3h] "s"((uint32_t)(P3 >> 32)), [p5l] "s"((uint32_t)P5), [p5h] "s"((uint32_t)(P5 >> 32)), \
      [p6l] "s"((uint32_t)P6), [p6h] "s"((uint32_t)(P6 >> 32)), [p4l] "s"((uint32_t)P4), [p4h] "s"((uint32_t)(P4 >> 32)),   \
      [seedl] "s"((uint32_t)seed), [seedh] "s"((uint32_t)(seed >> 32))
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
  static constexpr int WIN = K <= 21 ? 32 : 48;     // bases of a lane's code window (its M starts + K - 1 more): 64 / 96 bits
  static constexpr int LOOK_UNITS = (WIN - M + M - 1) / M;  // the last hashing lane's window ends inside the staged area: 2 / 3 units
  static constexpr int TILE = (WG - LOOK_UNITS) * M;  // 3 048 / 3 036 starts
  static constexpr int TILES = 9;
  static constexpr int ITEM = TILE * TILES;         // 27 432 / 27 324 starts per work item
  static constexpr int UNITS = WG;
  static constexpr int NB_T = UNITS * M;            // staged bases (3 072)
  static constexpr int N_R = NB_T + ((K + 3 - NB_T) & 3);  // length the reverse strand is indexed in: (N_R - K) & 3 == 3
  static constexpr int S = 4 * (NB_T / 4 + 3);      // bytes per phase image
  static constexpr int ND = (K + 3) / 4, NB = K - 4 * (ND - 1), NW = (K + 7) / 8;
  static_assert(M + K - 1 <= WIN && K <= 32, "a lane's k-mers live in its code window; 2 K bits fit a 64-bit compare");
  static_assert(((N_R - K) & 3) == 3, "reverse phase of start j is 3 - (j & 3)");
};

// PACKED: the genome is a hg_pack2 blob (include/hypergen.h: 2-bit codes, 4 bases per byte, then the not-a-base bitmap;
// 0.375 bytes per base in HBM instead of 1) and the tile level does no classification at all: a lane fetches its whole
// 32- / 48-base code window (the canonical compare's operand, as loaded) and the window's validity bits with two
// unaligned loads, and the forward / reverse-complement ASCII of four bases comes out of ONE 8-byte LDS table read
// indexed by the code byte (256 entries x {ASCII, complement ASCII byte-reversed}) -- the pattern of the reference's
// second kernel (src/cuda_kernel.cu:15-69, NT4 codes from src/sketch_cuda.rs:23-32), not its data path.
template <int K, bool CANON, bool PACKED>
__global__ __launch_bounds__(WG) void kmer_sample_shared(
    const uint8_t *__restrict__ seq, const hg_genome_meta *__restrict__ meta,
    const uint32_t *__restrict__ item_genome, uint64_t threshold, uint64_t seed, uint32_t u2t,
    uint64_t *__restrict__ hits, uint32_t *__restrict__ cnt, uint32_t stage_cap, const uint32_t *__restrict__ group_first) {
  using G = GeoS<K>;
  constexpr int M = G::M, DW = G::DW, S = G::S, ND = G::ND, NB = G::NB, NW = G::NW, N_R = G::N_R, WIN = G::WIN;
  typedef typename std::conditional<(WIN > 32), uint64_t, uint32_t>::type inv_t;  // validity bits of the code window
  constexpr inv_t MASKK = (inv_t)(((uint64_t)1 << K) - 1);

  // A workgroup takes the work items [item_lo, item_hi) one after the other: ONE item of a genome that fills it (TILES
  // tiles), or the items of several consecutive SMALL genomes, TILES tiles between them (the host's plan groups them:
  // hg_sketch_plan.hip) -- a genome of a few kbp is one or two tiles, and a workgroup of its own paid the launch, the
  // dependent loads of its records and the table set-up below for 6 us of hashing.  Everything that describes the current
  // item lives in the variables below; the staging lambdas read them by reference.
  const uint32_t tid = threadIdx.x;
  const uint32_t item_lo = group_first ? group_first[blockIdx.x] : blockIdx.x;
  const uint32_t item_hi = group_first ? group_first[blockIdx.x + 1] : blockIdx.x + 1;
  uint32_t g = 0;
  hg_genome_meta gm{};
  uint64_t n_bps = 0, n_starts = 0, item_start = 0;
  const uint8_t *__restrict__ gseq = seq;
  uint32_t tile_no = 0;  // tiles this workgroup has done (parity of the "dirty" flags)

  __shared__ HitStage stage;
  __shared__ __attribute__((aligned(16))) uint32_t s_f[4 * S / 4];                 // forward phase images
  __shared__ __attribute__((aligned(16))) uint32_t s_r[CANON ? 4 * S / 4 : 4];     // reverse phase images (psi at S (3 - psi))
  __shared__ __attribute__((aligned(16))) uint8_t s_code[(CANON && !PACKED) ? G::NB_T / 4 + 16 : 16];  // 2-bit codes, base b at bits 2b..2b+1
  __shared__ uint32_t s_val[PACKED ? 4 : G::UNITS + 4];   // per unit of M bases: bit b set <=> base b cannot be part of a k-mer
  __shared__ uint32_t s_dirty[2];            // "some base of this tile is not ACGT / lies behind the genome end", by tile parity
  __shared__ __attribute__((aligned(8))) uint2 s_lut[PACKED ? 256 : 1];  // code byte -> {ASCII of its 4 bases, complement ASCII byte-reversed}
  if (tid == 0) stage.n = 0, s_dirty[0] = 0, s_dirty[1] = 0;
  if constexpr (!PACKED) {
    if (tid < 4) s_val[G::UNITS + tid] = 0;
  } else {
    static_assert(WG == 256, "one table entry per lane");
    uint32_t f = 0, r = 0;
#pragma unroll
    for (int b = 0; b < 4; ++b) {
      const uint32_t cd = (tid >> (2 * b)) & 3u;
      f |= ((0x54474341u >> (8 * cd)) & 0xFFu) << (8 * b);        // "ACGT"[code]
      r |= ((0x41434754u >> (8 * cd)) & 0xFFu) << (8 * (3 - b));  // "TGCA"[code], base b at byte 3 - b
    }
    s_lut[tid] = make_uint2(f, r);
  }
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

  // PACKED staging of unit u: the lane's code window (c0 = bases [P, P + 16): its own M bases + the lookahead dword;
  // c1, c2 = the rest of its WIN-base window) and the window's validity bits straight from the blob.  Positions behind
  // the genome end are invalid whatever the blob's padding says.
  typedef uint32_t __attribute__((aligned(1))) u32u;
  typedef uint64_t __attribute__((aligned(1))) u64u;
  const uint8_t *__restrict__ gmask = seq;
  auto stage_unit_packed = [&](uint32_t u, uint64_t tile_start, uint32_t par, uint32_t &c0, uint32_t &c1, uint32_t &c2,
                               inv_t &invw) __attribute__((always_inline)) {
    const uint64_t P = tile_start + (uint64_t)u * M;  // a multiple of 4: whole code bytes
    const int64_t rem64 = (int64_t)n_bps - (int64_t)P;
    const uint64_t Pc = rem64 > 0 ? P : 0;            // (a window that starts behind the end reads the blob's first bytes: all invalid anyway)
    // the blob (codes padded to 16 bytes + >= 16 bytes of bitmap) and the 32 readable bytes the caller leaves behind it cover
    // every byte fetched here
    const uint8_t *cp = gseq + (Pc >> 2);
    c0 = *reinterpret_cast<const u32u *>(cp);
    c1 = *reinterpret_cast<const u32u *>(cp + 4);
    c2 = WIN > 32 ? *reinterpret_cast<const u32u *>(cp + 8) : 0u;
    const uint64_t mb = *reinterpret_cast<const u64u *>(gmask + (Pc >> 3));
    invw = (inv_t)(mb >> (uint32_t)(Pc & 7));
    if (rem64 < WIN) invw |= rem64 <= 0 ? ~(inv_t)0 : (inv_t)(~(inv_t)0 << (uint32_t)rem64);
    const bool dirty = invw != 0;
    if (__any(dirty)) {
      if ((threadIdx.x & 63) == 0 || dirty) s_dirty[par] = 1u;  // (same value from every writer)
    }
    uint32_t FA[DW + 1], RW[DW + 1];  // RW[t]: byte k = complement of base 4 t + 3 - k
#pragma unroll
    for (int t = 0; t <= DW; ++t) {
      const uint2 e = s_lut[(c0 >> (8 * t)) & 0xFFu];
      FA[t] = e.x, RW[t] = e.y;
    }
    uint32_t *const f0 = s_f + DW * u;
#pragma unroll
    for (int t = 0; t < DW; ++t) {
      f0[t] = FA[t];
      f0[S / 4 + t] = __builtin_amdgcn_alignbyte(FA[t + 1], FA[t], 1);
      f0[2 * (S / 4) + t] = __builtin_amdgcn_alignbyte(FA[t + 1], FA[t], 2);
      f0[3 * (S / 4) + t] = __builtin_amdgcn_alignbyte(FA[t + 1], FA[t], 3);
    }
    if constexpr (CANON) {
      // reverse phase images as in stage_unit: dword JB - (DW u + t) of phase psi holds the complements of bases
      // 4 t + ph + 3 .. 4 t + ph -- bytes [4 - ph, 8 - ph) of {RW[t] : RW[t + 1]}
#pragma unroll
      for (int ph = 0; ph < 4; ++ph) {
        const int psi = (N_R - ph) & 3, JB = (N_R - 4 - psi - ph) / 4;
        uint32_t *const r0 = s_r + (3 - psi) * (S / 4) + JB - (int)(DW * u);
#pragma unroll
        for (int t = 0; t < DW; ++t) r0[-t] = ph == 0 ? RW[t] : __builtin_amdgcn_alignbyte(RW[t], RW[t + 1], 4 - ph);
        if (ph > 0 && u == 0) r0[1] = __builtin_amdgcn_alignbyte(0u, RW[0], 4 - ph);
      }
    }
  };

  // (the next item's records are requested while this one is hashed: two dependent scalar loads otherwise in front of
  // every small genome)
  uint32_t g_next = item_genome[item_lo];
  hg_genome_meta gm_next = meta[g_next];
#pragma unroll 1
  for (uint32_t item = item_lo; item < item_hi; ++item) {
  g = g_next;
  gm = gm_next;
  if (item + 1 < item_hi) {
    g_next = item_genome[item + 1];
    gm_next = meta[g_next];
  }
  n_bps = gm.n_bps;
  if (n_bps < (uint64_t)K) continue;  // (uniform; such a genome has no work item anyway)
  n_starts = n_bps - K + 1;
  gseq = seq + gm.seq_off;
  gmask = seq + gm.mask_off;
  item_start = (uint64_t)(item - gm.item_first) * G::ITEM;
#pragma unroll 1
  for (int tile = 0; tile < G::TILES; ++tile, ++tile_no) {
    const uint64_t tile_start = item_start + (uint64_t)tile * G::TILE;
    if (tile_start >= n_starts) break;  // uniform
    const uint32_t par = tile_no & 1u;
    uint32_t pc0 = 0, pc1 = 0, pc2 = 0;
    inv_t pinv = 0;
    // (a wave whose units all lie behind everything a live lane's window can reach -- the last k-mer start + WIN bases -- stages
    // nothing: wave-uniform; its slots of the images keep what an earlier tile left there and nobody reads them)
    if (tile_start + (uint64_t)(tid & ~63u) * M < n_starts + (uint64_t)WIN) {
      if constexpr (PACKED) stage_unit_packed(tid, tile_start, par, pc0, pc1, pc2, pinv);
      else stage_unit(tid, tile_start, par);
    }
#ifndef HG_KS_EXP
#define HG_KS_EXP 0
#endif
    if (!(HG_KS_EXP & 1)) __syncthreads();

    const bool tile_dirty = s_dirty[par] != 0u;  // workgroup-uniform
    inv_t inv_w = 0;
    if constexpr (PACKED) {
      if (tile_dirty) inv_w = pinv;
    } else if (tile_dirty) {
      if constexpr (WIN > 32)
        inv_w = (inv_t)((uint64_t)s_val[tid] | ((uint64_t)s_val[tid + 1] << M) | ((uint64_t)s_val[tid + 2] << (2 * M)) |
                        ((uint64_t)s_val[tid + 3] << (3 * M)));
      else
        inv_w = (inv_t)(s_val[tid] | (s_val[tid + 1] << M) | (s_val[tid + 2] << (2 * M)));
    }
    // the lane's code window: WIN bases from bit 2 M tid = byte (M / 4) tid of the code image.  LSB-first complement
    // stream (read upwards it IS the reverse strand) and MSB-first forward stream (base 0 in the top two bits), as
    // 64-bit values (WIN = 32) or dwords (WIN = 48)
    uint64_t Gm = 0, Gc = 0;
    uint32_t gf[3] = {0, 0, 0}, wc[3] = {0, 0, 0};
    if constexpr (CANON) {
      uint32_t wl, wh, w2 = 0;
      if constexpr (PACKED) {
        wl = pc0, wh = pc1, w2 = pc2;  // the window as loaded
      } else {
        const uint32_t bo = (uint32_t)DW * tid;
        const uint32_t *cw = reinterpret_cast<const uint32_t *>(s_code) + (bo >> 2);
        const uint32_t c0 = cw[0], c1 = cw[1], c2 = cw[2], sh = 8u * (bo & 3u);
        wl = __builtin_amdgcn_alignbit(c1, c0, sh), wh = __builtin_amdgcn_alignbit(c2, c1, sh);
        if constexpr (WIN > 32) w2 = __builtin_amdgcn_alignbit(cw[3], c2, sh);
      }
      auto pairrev = [](uint32_t v) {
        const uint32_t br = __builtin_bitreverse32(v);
        return ((br >> 1) & 0x55555555u) | ((br & 0x55555555u) << 1);
      };
      if constexpr (WIN == 32) {
        Gc = ~mk64(wl, wh);
        Gm = mk64(pairrev(wh), pairrev(wl));
      } else {
        wc[0] = ~wl, wc[1] = ~wh, wc[2] = ~w2;
        gf[0] = pairrev(wl), gf[1] = pairrev(wh), gf[2] = pairrev(w2);
      }
    }

    // the chosen strand's base address for k-mer jj: forward < reverse complement as one 64-bit compare of 2-bit codes
    auto strand_base = [&](auto jjc) __attribute__((always_inline)) -> uint32_t {
      constexpr int jj = decltype(jjc)::value;
      uint32_t base = aF;
      if constexpr (CANON) {
        uint64_t fv, rv;
        if constexpr (WIN == 32) {
          if constexpr ((K & 1) != 0) {
            // odd K: the compare is decided inside the 2K bits, the values only have to be TOP-aligned
            fv = Gm << (2 * jj);
            rv = Gc << (2 * (32 - K - jj));
          } else {
            constexpr uint64_t MASK2K = (1ull << (2 * K)) - 1;
            fv = (Gm >> (2 * (32 - K - jj))) & MASK2K;
            rv = (Gc >> (2 * jj)) & MASK2K;
          }
        } else {
          // 96-bit streams: both values TOP-aligned in 64 bits.  Forward: the MSB-first stream from bit 2 jj (from the
          // top).  Reverse: bits [e, e + 64) of the LSB-first complement stream, e = 2 jj + 2 K - 64 (zeros below
          // bit 0).  Even K: what lies below the 2 K bits is cut off; odd K: it cannot decide.
          constexpr int o = 2 * jj;
          uint32_t fh, fl;
          if constexpr (o == 0) fh = gf[0], fl = gf[1];
          else fh = __builtin_amdgcn_alignbit(gf[0], gf[1], 32 - o), fl = __builtin_amdgcn_alignbit(gf[1], gf[2], 32 - o);
          constexpr int e = 2 * jj + 2 * K - 64, off = e + 32, a = off >> 5, sft = off & 31;
          static_assert(off >= 0 && a <= 1, "window inside [0, wc0, wc1, wc2]");
          const uint32_t x_[4] = {0u, wc[0], wc[1], wc[2]};
          uint32_t rl, rh;
          if constexpr (sft == 0) rl = x_[a], rh = x_[a + 1];
          else rl = __builtin_amdgcn_alignbit(x_[a + 1], x_[a], sft), rh = __builtin_amdgcn_alignbit(x_[a + 2], x_[a + 1], sft);
          if constexpr ((K & 1) == 0 && K < 32) {
            constexpr uint32_t LOWCUT = ~((1u << (64 - 2 * K)) - 1u);
            fl &= LOWCUT, rl &= LOWCUT;
          }
          fv = mk64(fl, fh), rv = mk64(rl, rh);
        }
        uint64_t lt;
        asm("v_cmp_lt_u64_e64 %0, %1, %2" : "=s"(lt) : "v"(rv), "v"(fv));
        asm("v_cndmask_b32_e64 %0, %1, %2, %3" : "=v"(base) : "v"(aF), "v"(aR[jj >> 2]), "s"(lt));
      }
      return base;
    };
    auto fetch_words = [&](auto jjc, uint64_t *w) __attribute__((always_inline)) {
      constexpr int jj = decltype(jjc)::value;
      const uint32_t base = strand_base(jjc);
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
    constexpr bool ASM_BODY = HG_KS_ASM && (NW == 3 || NW == 4);  // K = 17..32
    // One asm statement per k-mer: wait for this k-mer's words (requested a whole hash earlier), request the next
    // k-mer's into the other buffer (whole dwords of the chosen strand's phase image: two 8-byte reads at 4-byte aligned
    // addresses, one dword and the last 1..4 bytes), hash, compare with the threshold.  The buffers are bound to fixed
    // registers on both sides, so the compiler sees ordinary values and never copies them.
    uint64_t w0a = 0, w0b = 0, w0c = 0, w0d = 0, w1a = 0, w1b = 0, w1c = 0, w1d = 0;  // parity 0: v[56:61] + v[90:91], parity 1: v[62:67] + v[92:93]
    uint64_t hmask = 0, hjunk = 0, hval = 0;
    uint64_t zpair = 0;  // Z = {x, 0}: the zero-extension pair of the products' high dwords; every asm statement rewrites
                         // its low half only, so the zero in the high half is carried from k-mer to k-mer as a value
    constexpr int KCASE = 10 * ND + NB;
#define HG_KS_OFFS [o0] "n"(IMM), [o1] "n"(IMM + 8), [o2] "n"(IMM + 16), [o3] "n"(IMM + 20), [o4] "n"(IMM + 24), [o5] "n"(IMM + 28)
#define HG_KS_OUT0 "={v[56:57]}"(w0a), "={v[58:59]}"(w0b), "={v[60:61]}"(w0c), "={v[90:91]}"(w0d)
#define HG_KS_OUT1 "={v[62:63]}"(w1a), "={v[64:65]}"(w1b), "={v[66:67]}"(w1c), "={v[92:93]}"(w1d)
#define HG_KS_IN0 "{v[56:57]}"(w0a), "{v[58:59]}"(w0b), "{v[60:61]}"(w0c), "{v[90:91]}"(w0d)
#define HG_KS_IN1 "{v[62:63]}"(w1a), "{v[64:65]}"(w1b), "{v[66:67]}"(w1c), "{v[92:93]}"(w1d)
#define HG_KS_FIRST(C, N)                                                                                             \
  if constexpr (KCASE == C)                                                                                           \
    asm volatile(HG_KS_APPLY(HG_KS_READ_TEXT_##C, HG_KS_REGS0) : HG_KS_OUT0 : [base] "v"(base), HG_KS_OFFS);
#define HG_KS_EVEN(C, N)                                                                                              \
  if constexpr (KCASE == C)                                                                                           \
    asm volatile(HG_KS_APPLY(HG_KS_WAIT_TEXT_##C, HG_KS_REGS0) HG_KS_APPLY(HG_KS_READ_TEXT_##C, HG_KS_REGS1) "\n\t"   \
                 HG_KS_APPLY(HG_KS_HASH_TEXT_##N, HG_KS_PAIRS0)                                                       \
                 : [mask] "=s"(hmask), [junk] "=&s"(hjunk), "={v[70:71]}"(hval), "={v[68:69]}"(zpair), HG_KS_OUT1     \
                 : HG_KS_HASH_INPUTS, "{v[68:69]}"(zpair), [base] "v"(base), HG_KS_OFFS, HG_KS_IN0                    \
                 : HG_KS_HASH_CLOBBERS);
#define HG_KS_ODD(C, N)                                                                                               \
  if constexpr (KCASE == C)                                                                                           \
    asm volatile(HG_KS_APPLY(HG_KS_WAIT_TEXT_##C, HG_KS_REGS1) HG_KS_APPLY(HG_KS_READ_TEXT_##C, HG_KS_REGS0) "\n\t"   \
                 HG_KS_APPLY(HG_KS_HASH_TEXT_##N, HG_KS_PAIRS1)                                                       \
                 : [mask] "=s"(hmask), [junk] "=&s"(hjunk), "={v[70:71]}"(hval), "={v[68:69]}"(zpair), HG_KS_OUT0     \
                 : HG_KS_HASH_INPUTS, "{v[68:69]}"(zpair), [base] "v"(base), HG_KS_OFFS, HG_KS_IN1                    \
                 : HG_KS_HASH_CLOBBERS);
#define HG_KS_LAST(C, N)                                                                                              \
  if constexpr (KCASE == C)                                                                                           \
    asm volatile(HG_KS_APPLY(HG_KS_WAIT_TEXT_##C, HG_KS_REGS1) HG_KS_APPLY(HG_KS_HASH_TEXT_##N, HG_KS_PAIRS1)         \
                 : [mask] "=s"(hmask), [junk] "=&s"(hjunk), "={v[70:71]}"(hval), "={v[68:69]}"(zpair)                 \
                 : HG_KS_HASH_INPUTS, "{v[68:69]}"(zpair), HG_KS_IN1                                                  \
                 : HG_KS_HASH_CLOBBERS);
#define HG_KS_ALL_CASES(X) X(51, 3) X(52, 3) X(53, 3) X(54, 3) X(61, 3) X(62, 3) X(63, 3) X(64, 3) \
                           X(71, 4) X(72, 4) X(73, 4) X(74, 4) X(81, 4) X(82, 4) X(83, 4) X(84, 4)
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
          const bool valid = !CHECK || ((inv_w >> j) & MASKK) == 0;
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
            if (below && valid && hashing) stage_hit(stage, stage_cap, hval, gm, g, hits, cnt);
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
        const bool valid = !CHECK || ((inv_w >> j) & MASKK) == 0;
        if constexpr (j == 0) fetch_words(jc, wq[0]);
        if constexpr (j + 1 < M) fetch_words(std::integral_constant<int, j + 1>{}, wq[(j + 1) & 1]);
        const uint64_t h = t1ha2_fixed_w<K>(wq[j & 1], seed);
        if (valid && hashing && h < threshold) stage_hit(stage, stage_cap, h, gm, g, hits, cnt);
      });
    };
    // A wave whose 64 x M starts all lie behind the genome's last k-mer hashes nothing (wave-uniform): a 2 kbp genome fills
    // three of its one tile's four waves, the last tile of a 10 kbp genome two -- the idle wave only meets the others at the
    // barriers, and its issue slots go to the other workgroups of the CU (2 kbp genomes: 0.27 -> 0.4 of the per-base rate).
    const bool wave_live = tile_start + (uint64_t)(tid & ~63u) * M < n_starts;
    if (wave_live) {
      if constexpr (ASM_BODY) {
        if (!tile_dirty) kmers_asm(std::false_type{});
        else kmers_asm(std::true_type{});
      } else {
        if (!tile_dirty) run_kmers(std::false_type{});
        else run_kmers(std::true_type{});
      }
    }
    if (!(HG_KS_EXP & 1)) __syncthreads();  // every read of the images is done: the next tile may overwrite them
    if (tid == 0) s_dirty[par] = 0u;  // (raised again in two tiles' time at the earliest, behind the next tile's barriers)
    // (flush_hits_if_filling branches on the staged count, workgroup-uniform only behind the barrier above: a development
    // build that compiles that barrier out puts its own in front)
    if (dense_sampling(threshold)) {
      if (HG_KS_EXP & 1) __syncthreads();
      flush_hits_if_filling(stage, stage_cap, gm, g, hits, cnt);
    }
  }
  // the item's hits to its genome's region (the list is empty again for the next item of the group: its first tile's
  // barrier lies between this call's reads of the list and the next writes)
  if (item + 1 < item_hi) flush_hits<true>(stage, stage_cap, gm, g, hits, cnt);
  else flush_hits(stage, stage_cap, gm, g, hits, cnt);
  }
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
// A workgroup stages 2 048 bytes of sequence per tile -- 1 536 k-mer starts and the up to 254 bytes behind the last one --
// into LDS as the normalised forward strand (upper-case ASCII, 0 for anything that is not a base) and its reverse
// complement (so that the reverse strand of a k-mer is an ascending byte range too), each in all four byte phases (image
// phi holds bytes [4 i + phi, 4 i + phi + 4) at dword i: every dword of a k-mer that starts at ANY byte is one aligned read
// of the image of its start's phase -- the pattern of kmer_sample_shared), plus one validity bit per byte.  A lane stages
// eight bytes: it loads them with the dword in front and the dword behind (16 bytes), classifies all four dwords, and
// writes its two dwords of all eight images itself -- the forward phases need the NEXT dword, the reverse-complement phases
// the PREVIOUS one -- so a tile has two barriers and no pass over LDS.  (Until round 6: 1 024 starts per tile, bytes stored
// one by one, a zero fill, an invalid-byte prefix scan and a second pass for phases 1..3 -- five barriers, and half of the
// kernel's 25 ms for 1 000 x 5 Mbp at k = 33 was spent before any hashing.)
// Each lane then takes six starts.  KC = 0: run-time k (65..255).  KC = 33..64: k is a compile-time constant -- the validity
// test is two funnel shifts on the bit image, the strand choice one 64-bit compare of the first eight bases (the byte
// strings differ there for all but one k-mer in 65 536), and t1ha2 is straight-line code: one round of its 32-byte loop
// (two for k = 64) and a tail of k - 32 bytes.
constexpr uint32_t LONG_TILE = 1536;                 // k-mer starts per tile (6 per lane)
constexpr uint32_t LONG_BYTES = 2048;                // staged bytes per tile (>= LONG_TILE + 254), 8 per lane
// dwords between two phase images: the tile + slack.  Consecutive lanes take consecutive starts, so a wave reads 16
// consecutive dwords of each of the four images at once; with the images 16 banks apart (pitch == 16 mod 32) every LDS bank
// serves exactly two lanes, with 516 dwords (pitch 4) up to four.  Measured at k = 33: 18.4 against 18.1 ms -- no gain, the
// kernel is VALU-bound (SQ_LDS_BANK_CONFLICT is 61 % of its LDS cycles either way, the LDS is busy 59 % of the time); the
// switches HG_LK_PITCH / HG_LK_DIRTY / HG_LK_CACHE are the A/B builds of tools/build_variant.sh.
#ifndef HG_LK_PITCH
#define HG_LK_PITCH 16
#endif
constexpr uint32_t LONG_DW = LONG_BYTES / 4 + HG_LK_PITCH;
static_assert(GEN_ITEM % LONG_TILE == 0 && LONG_TILE % 8 == 0 && LONG_BYTES == 8 * WG && LONG_TILE + 254 <= LONG_BYTES, "long-k tile geometry");

struct LdsStrand {
  const uint32_t *base;  // dword 0 of the k-mer in the phase image of its start (image (byte0 & 3), dword byte0 >> 2)
  uint32_t d0, d1;       // its first two dwords when `cached` (the strand choice has read them already)
  bool cached;
  __device__ __forceinline__ uint32_t dword(uint32_t i) const {  // bytes [byte0 + 4i, byte0 + 4i + 4)
    if (cached && __builtin_constant_p(i) && i == 0) return d0;
    if (cached && __builtin_constant_p(i) && i == 1) return d1;
    return base[i];
  }
  __device__ __forceinline__ uint64_t word(uint32_t byte_off, uint32_t nbytes) const {  // little endian, byte_off % 8 == 0
    uint32_t lo = dword(byte_off / 4), hi = nbytes > 4 ? dword(byte_off / 4 + 1) : 0u;
    if (nbytes < 4) lo &= (1u << (8 * nbytes)) - 1;
    else if (nbytes > 4 && nbytes < 8) hi &= (1u << (8 * (nbytes - 4))) - 1;
    return mk64(lo, hi);
  }
};

// rot64 whose result is opaque to the optimiser.  `x + rot64(y, s)`: the compiler sees the rotate as {lo} | {hi << 32}, turns
// the or of disjoint halves into an add, and adds the two halves SEPARATELY -- two 64-bit adds and two v_mov (zero halves) per
// rotate where one add would do; eight rotates per k-mer in the long-input round.  The empty asm makes the pair one value.
__device__ __forceinline__ uint64_t rot64p(uint64_t v, unsigned s) {
  uint64_t r = rot64(v, s);
  asm("" : "+v"(r));
  return r;
}

// t1ha2_atonce of `len` > 32 bytes (published t1ha2: lanes c, d, 32 bytes per round, squash, then the tail switch of
// src/cuda_kernel.cu:207-245).  LEN > 0: len == LEN is a compile-time constant and everything below unrolls.
template <uint32_t LEN>
__device__ __forceinline__ uint64_t t1ha2_long(const LdsStrand &sb, uint32_t len_rt, uint64_t seed) {
  const uint32_t len0 = LEN ? LEN : len_rt;
  uint64_t a = seed, b = (uint64_t)len0;
  uint32_t off = 0, len = len0;
  {
    uint64_t c = rot64((uint64_t)len, 23) + ~seed;
    uint64_t d = ~(uint64_t)len + rot64(seed, 19);
    auto round = [&]() __attribute__((always_inline)) {
      const uint64_t w0 = sb.word(off, 8), w1 = sb.word(off + 8, 8), w2 = sb.word(off + 16, 8), w3 = sb.word(off + 24, 8);
      off += 32;
      const uint64_t d02 = w0 + rot64p(w2 + d, 56);
      const uint64_t c13 = w1 + rot64p(w3 + c, 19);
      d ^= b + rot64p(w1, 38);
      c ^= a + rot64p(w0, 57);
      b ^= lo64mul<P6>(c13 + w2);
      a ^= lo64mul<P5>(d02 + w3);
    };
    if constexpr (LEN != 0) {
      static_assert(LEN > 32 && LEN <= 64, "compile-time lengths: one or two rounds");
      round();
      if constexpr (LEN == 64) round();  // (data < detent  <=>  off + 31 < len)
    } else {
      do round();
      while (off + 31 < len);
    }
    a ^= lo64mul<P6>(c + rot64p(d, 23));
    b ^= lo64mul<P5>(rot64p(c, 19) + d);
    len &= 31;
  }
  uint32_t rem = len;
  if (rem > 24) { mixup64<P4>(a, b, sb.word(off, 8)); off += 8, rem -= 8; }
  if (rem > 16) { mixup64<P3>(b, a, sb.word(off, 8)); off += 8, rem -= 8; }
  if (rem > 8) { mixup64<P2>(a, b, sb.word(off, 8)); off += 8, rem -= 8; }
  if (rem > 0) mixup64<P1>(b, a, sb.word(off, rem));
  return final64(a, b);
}

// four bases as ASCII -> {normalised forward ASCII, complement ASCII, invalid bytes as 0xFF}; bytes that are not a base
// (after the optional u/U -> T) are 0 in both strands.  The classification of kmer_sample_shared's stage_unit.
__device__ __forceinline__ void long_classify_ascii(uint32_t x, uint32_t u2t, uint32_t &fa, uint32_t &ca, uint32_t &bad) {
  if (u2t) {
    const uint32_t e = (x & 0xDFDFDFDFu) ^ 0x55555555u;
    const uint32_t nz = ((e & 0x7F7F7F7Fu) + 0x7F7F7F7Fu) | e;  // bit 7 set <=> byte != 'U'
    x ^= (~nz & 0x80808080u) >> 7;                               // 'U' ^ 'T' == 1
  }
  const uint32_t tt = x ^ (x >> 1);
  const uint32_t cd = (tt >> 1) & 0x03030303u;  // A,C,G,T -> 0,1,2,3 per byte
  const uint32_t f = __builtin_amdgcn_perm(0u, 0x54474341u, cd), c = __builtin_amdgcn_perm(0u, 0x41434754u, cd);
  const uint32_t dif = (x & 0xDFDFDFDFu) ^ f;
  const uint32_t z = (((dif & 0x7F7F7F7Fu) + 0x7F7F7F7Fu) | dif) & 0x80808080u;  // bit 7 of a byte set <=> not a base
  bad = (z >> 7) * 0xFFu;
  fa = f & ~bad, ca = c & ~bad;
}
// four bases as 2-bit codes (low byte of `codes`) + their four not-a-base bits
__device__ __forceinline__ void long_classify_codes(uint32_t codes, uint32_t inv4, uint32_t &fa, uint32_t &ca, uint32_t &bad) {
  const uint32_t cd = (codes & 3u) | ((codes & 0xCu) << 6) | ((codes & 0x30u) << 12) | ((codes & 0xC0u) << 18);
  bad = (((inv4 & 15u) * 0x00204081u) & 0x01010101u) * 0xFFu;
  fa = __builtin_amdgcn_perm(0u, 0x54474341u, cd) & ~bad;
  ca = __builtin_amdgcn_perm(0u, 0x41434754u, cd) & ~bad;
}

template <int KC, bool PACKED>  // (PACKED last: a profiler's name of a packed-input kernel ends in "true>" for both kernels)
__global__ __launch_bounds__(WG) void kmer_sample_long(
    const uint8_t *__restrict__ seq, const hg_genome_meta *__restrict__ meta,
    const uint32_t *__restrict__ item_genome, uint32_t ksize_rt, uint64_t threshold, uint64_t seed,
    uint32_t canonical, uint32_t u2t, uint64_t *__restrict__ hits, uint32_t *__restrict__ cnt, uint32_t stage_cap) {
  __shared__ HitStage stage;
  __shared__ uint32_t s_f[4 * LONG_DW], s_rc[4 * LONG_DW];  // forward / reverse-complement strand, four byte phases each
  __shared__ uint32_t s_inv[LONG_BYTES / 32 + 12];          // bit i set <=> staged byte i is not a base (zero slack behind)
  __shared__ uint32_t s_anybad[2];                          // "this tile holds a byte that is not a base", by tile parity
  const uint32_t ksize = KC ? (uint32_t)KC : ksize_rt;
  const uint32_t item = blockIdx.x, tid = threadIdx.x;
  const uint32_t g = item_genome[item];
  const hg_genome_meta gm = meta[g];
  const uint64_t n_bps = gm.n_bps;
  if (n_bps < ksize) return;
  const uint64_t n_starts = n_bps - ksize + 1;
  const uint8_t *__restrict__ gseq = seq + gm.seq_off;
  const uint8_t *__restrict__ gmask = seq + gm.mask_off;  // PACKED: the genome's not-a-base bitmap
  const uint64_t item_start = (uint64_t)(item - gm.item_first) * GEN_ITEM;
  if (tid == 0) stage.n = 0;
  if (tid < 4) s_f[tid * LONG_DW + LONG_BYTES / 4] = 0u, s_rc[tid * LONG_DW + LONG_BYTES / 4] = 0u;  // (read by a k-mer's last, masked dword)
  if (tid < 12) s_inv[LONG_BYTES / 32 + tid] = 0u;
  if (tid < 2) s_anybad[tid] = 0u;
  typedef uint32_t __attribute__((aligned(1))) u32u_t;
  uint32_t tile_par = 0;

  for (uint64_t tile0 = item_start; tile0 < item_start + GEN_ITEM && tile0 < n_starts; tile0 += LONG_TILE, tile_par ^= 1u) {
    __syncthreads();  // previous tile's readers are done
    if (tid == 0) s_anybad[tile_par ^ 1u] = 0u;  // (the flag of the tile before: read by everyone in front of the barrier above)
    if (dense_sampling(threshold)) flush_hits_if_filling(stage, stage_cap, gm, g, hits, cnt);
    // ---- stage: the lane's eight bytes [pos0, pos0 + 8) with the four bytes in front of and behind them
    {
      const uint64_t pos0 = tile0 + (uint64_t)tid * 8;  // a multiple of 8
      uint32_t fa[4], ca[4], bad[4];                     // dwords at pos0 - 4, pos0, pos0 + 4, pos0 + 8
#pragma unroll
      for (int t = 0; t < 4; ++t) {
        const int64_t q = (int64_t)pos0 + 4 * t - 4;    // first base of this dword
        fa[t] = ca[t] = 0u, bad[t] = ~0u;
        if (q >= 0 && (uint64_t)q < n_bps) {             // (the caller leaves 32 readable bytes behind every genome)
          if constexpr (PACKED) {
            const uint32_t codes = gseq[(uint64_t)q >> 2];  // q is a multiple of 4: one code byte
            const uint32_t inv4 = (uint32_t)gmask[(uint64_t)q >> 3] >> ((uint32_t)q & 4u);
            long_classify_codes(codes, inv4, fa[t], ca[t], bad[t]);
          } else {
            long_classify_ascii(*reinterpret_cast<const u32u_t *>(gseq + q), u2t, fa[t], ca[t], bad[t]);
          }
          const uint64_t left = n_bps - (uint64_t)q;     // bases of the genome from q on: what lies behind them is not sequence
          if (left < 4) {
            const uint32_t keep = (1u << (8 * (uint32_t)left)) - 1u;
            fa[t] &= keep, ca[t] &= keep, bad[t] |= ~keep;
          }
        }
      }
      // forward images: dwords 2 tid and 2 tid + 1 of every phase
#pragma unroll
      for (uint32_t ph = 0; ph < 4; ++ph) {
        s_f[ph * LONG_DW + 2 * tid] = ph ? __builtin_amdgcn_alignbyte(fa[2], fa[1], ph) : fa[1];
        s_f[ph * LONG_DW + 2 * tid + 1] = ph ? __builtin_amdgcn_alignbyte(fa[3], fa[2], ph) : fa[2];
      }
      // reverse complement: byte LONG_BYTES - 1 - i holds the complement of forward byte i, so forward dword f is dword
      // LONG_BYTES / 4 - 1 - f byte-reversed, and its successor in the image is the forward dword IN FRONT of it
      const uint32_t r0 = __builtin_bswap32(ca[0]), r1 = __builtin_bswap32(ca[1]), r2 = __builtin_bswap32(ca[2]);
      const uint32_t j1 = LONG_BYTES / 4 - 1 - 2 * tid;  // image dword of forward dword 2 tid (j1 - 1: of 2 tid + 1)
#pragma unroll
      for (uint32_t ph = 0; ph < 4; ++ph) {
        s_rc[ph * LONG_DW + j1] = ph ? __builtin_amdgcn_alignbyte(r0, r1, ph) : r1;
        s_rc[ph * LONG_DW + j1 - 1] = ph ? __builtin_amdgcn_alignbyte(r1, r2, ph) : r2;
      }
      // validity bits of the lane's eight bytes: byte tid of the bit image
      const uint32_t b8 = ((((bad[1] & 0x01010101u) * 0x01020408u) >> 24) & 15u) | (((((bad[2] & 0x01010101u) * 0x01020408u) >> 24) & 15u) << 4);
      reinterpret_cast<uint8_t *>(s_inv)[tid] = (uint8_t)b8;
      if (b8 != 0u) s_anybad[tile_par] = 1u;  // (same value from every writer)
    }
    __syncthreads();
#ifndef HG_LK_DIRTY
#define HG_LK_DIRTY 1
#endif
    const bool tile_dirty = !HG_LK_DIRTY || s_anybad[tile_par] != 0u;  // workgroup-uniform: a clean tile (nearly all of them) tests no window
    // ---- the lane's six starts p = tid + 256 j.  256 is a multiple of 4, so the byte phase of a lane's starts is the same for
    // all j on both strands and the k-mer's first dword moves by 64 dwords per j: two pointers that step, no address arithmetic
    // per start (forward: byte p; reverse complement: byte LONG_BYTES - p - k)
    const uint64_t left_starts = n_starts - tile0;  // > 0
    const uint32_t lim = left_starts < LONG_TILE ? (uint32_t)left_starts : LONG_TILE;
    const uint32_t jn = lim > tid ? (lim - tid + WG - 1) / WG : 0u;  // this lane's starts inside the genome
    const uint32_t rb0 = LONG_BYTES - ksize - tid;
    const uint32_t *fp = s_f + (tid & 3u) * LONG_DW + (tid >> 2);
    const uint32_t *rp = s_rc + (rb0 & 3u) * LONG_DW + (rb0 >> 2);
#ifndef HG_LK_UNROLL
#define HG_LK_UNROLL 1
#endif
#pragma unroll HG_LK_UNROLL
    for (uint32_t j = 0; j < jn; ++j, fp += WG / 4, rp -= WG / 4) {
      if (tile_dirty) {  // a non-base inside the window [p, p + k)?
        const uint32_t p = tid + WG * j, dw = p >> 5, sh = p & 31u;
        bool any_bad;
        if constexpr (KC != 0) {
          const uint32_t d0 = s_inv[dw], d1 = s_inv[dw + 1], d2 = s_inv[dw + 2];
          const uint32_t w0 = __builtin_amdgcn_alignbit(d1, d0, sh), w1 = __builtin_amdgcn_alignbit(d2, d1, sh);  // bits p .. p + 63
          constexpr uint32_t M1 = KC >= 64 ? ~0u : (1u << (KC - 32)) - 1u;
          any_bad = (w0 | (w1 & M1)) != 0u;
        } else {
          uint32_t acc = 0;
          for (uint32_t t = 0; 32 * t < ksize; ++t) {
            const uint32_t w = __builtin_amdgcn_alignbit(s_inv[dw + t + 1], s_inv[dw + t], sh);
            const uint32_t left = ksize - 32 * t;
            acc |= left >= 32 ? w : w & ((1u << left) - 1u);
          }
          any_bad = acc != 0u;
        }
        if (any_bad) continue;
      }
      // the lexicographically smaller byte string (canonical): big-endian compare of the first eight bytes, then the rest; the
      // forward strand's first two dwords are read in any case -- they are the hash's first word (plain values: a struct that
      // is written after its construction goes to scratch)
      const uint32_t fd0 = fp[0], fd1 = fp[1];
      uint32_t rd0 = 0, rd1 = 0;
      bool use_rc = false;
      if (canonical) {
        rd0 = rp[0], rd1 = rp[1];
        const uint64_t fw = mk64(__builtin_bswap32(fd1), __builtin_bswap32(fd0));
        const uint64_t rw = mk64(__builtin_bswap32(rd1), __builtin_bswap32(rd0));
        if (fw != rw) {
          use_rc = rw < fw;
        } else {
          for (uint32_t t = 2; 4 * t < ksize; ++t) {
            uint32_t fd = __builtin_bswap32(fp[t]), rd = __builtin_bswap32(rp[t]);
            const uint32_t left = ksize - 4 * t;
            if (left < 4) fd &= ~0u << (8 * (4 - left)), rd &= ~0u << (8 * (4 - left));
            if (fd != rd) {
              use_rc = rd < fd;
              break;
            }
          }
        }
      }
#ifndef HG_LK_CACHE
#define HG_LK_CACHE 1
#endif
      const LdsStrand sel{use_rc ? rp : fp, use_rc ? rd0 : fd0, use_rc ? rd1 : fd1, HG_LK_CACHE != 0};
      const uint64_t h = t1ha2_long<(uint32_t)KC>(sel, ksize, seed);
      if (h < threshold) stage_hit(stage, stage_cap, h, gm, g, hits, cnt);
    }
  }
  flush_hits(stage, stage_cap, gm, g, hits, cnt);
}

}  // namespace

const char *hg_kmer_kernel_name(uint32_t k, bool canonical, bool packed) {  // mirrors hg_launch_kmer_sample
  static thread_local char buf[64];
  if (k > 32) {
    snprintf(buf, sizeof buf, "kmer_sample_long<%u, %s>", k <= 64 ? k : 0u, packed ? "true" : "false");
    return buf;
  }
  snprintf(buf, sizeof buf, "kmer_sample_shared<%u, %s, %s>", k, canonical ? "true" : "false", packed ? "true" : "false");
  return buf;
}

uint32_t hg_kmer_item_starts(uint32_t k) {
  if (k > 32) return GEN_ITEM;
  return k >= 22 ? (uint32_t)GeoS<32>::ITEM : (uint32_t)GeoS<21>::ITEM;  // (the same within each code-window class)
}

uint32_t hg_kmer_tile_starts(uint32_t k) {
  if (k > 32) return 0;  // (kmer_sample_long takes one item per workgroup)
  return k >= 22 ? (uint32_t)GeoS<32>::TILE : (uint32_t)GeoS<21>::TILE;
}
// tiles a workgroup takes at most when the plan groups the work items of small genomes: three full items' worth (27 genomes of
// up to 3 kbp, 6 of 10 kbp; an item that fills its TILES tiles alone still has a workgroup of its own)
uint32_t hg_kmer_item_tiles(uint32_t k) { return k > 32 ? 0u : 3u * (uint32_t)GeoS<21>::TILES; }

hipError_t hg_launch_kmer_sample(hipStream_t st, const uint8_t *d_seq, const hg_genome_meta *d_meta,
                                 const uint32_t *d_item_genome, uint32_t n_items, uint32_t ksize,
                                 uint64_t threshold, uint64_t seed, bool canonical, uint32_t norm_mode,
                                 uint64_t *d_hits, uint32_t *d_cnt, bool packed, const uint32_t *d_group_first,
                                 uint32_t n_groups) {
  if (n_items == 0) return hipSuccess;
  if (ksize > 32 || n_groups == 0) d_group_first = nullptr;
  const uint32_t n_wg = d_group_first ? n_groups : n_items;
  const uint32_t u2t = (norm_mode == HG_NORM_U2T && !packed) ? 1u : 0u;  // (a blob was normalised when it was packed)
  // entries of a work item's LDS hit list: twice what one tile is expected to sample (+ slack), 256 .. 4 096
  auto stage_entries = [&](uint32_t tile_starts) {
    const double per_tile = (double)tile_starts * ((double)threshold / 18446744073709551616.0);
    uint32_t cap = HIT_STAGE;
    while (cap < HIT_STAGE_MAX && (double)cap < 2.0 * per_tile + 128.0) cap <<= 1;
    return cap;
  };
  const uint32_t cap_s = stage_entries((uint32_t)GeoS<21>::TILE), cap_l = stage_entries(LONG_TILE);
#define HG_K_LAUNCH(KK, CN, PK)                                                                            \
  hipLaunchKernelGGL((kmer_sample_shared<KK, CN, PK>), dim3(n_wg), dim3(WG), cap_s * sizeof(uint64_t), st, d_seq, d_meta, \
                     d_item_genome, threshold, seed, u2t, d_hits, d_cnt, cap_s, d_group_first)
#define HG_K_CASE(KK)                                                                                     \
  case KK:                                                                                                \
    if (canonical && packed) HG_K_LAUNCH(KK, true, true);                                                 \
    else if (canonical) HG_K_LAUNCH(KK, true, false);                                                     \
    else if (packed) HG_K_LAUNCH(KK, false, true);                                                        \
    else HG_K_LAUNCH(KK, false, false);                                                                   \
    return hipGetLastError();
  switch (ksize) {
    HG_K_CASE(1) HG_K_CASE(2) HG_K_CASE(3) HG_K_CASE(4) HG_K_CASE(5) HG_K_CASE(6) HG_K_CASE(7) HG_K_CASE(8)
    HG_K_CASE(9) HG_K_CASE(10) HG_K_CASE(11) HG_K_CASE(12) HG_K_CASE(13) HG_K_CASE(14) HG_K_CASE(15) HG_K_CASE(16)
    HG_K_CASE(17) HG_K_CASE(18) HG_K_CASE(19) HG_K_CASE(20) HG_K_CASE(21) HG_K_CASE(22) HG_K_CASE(23) HG_K_CASE(24)
    HG_K_CASE(25) HG_K_CASE(26) HG_K_CASE(27) HG_K_CASE(28) HG_K_CASE(29) HG_K_CASE(30) HG_K_CASE(31) HG_K_CASE(32)
    default:
      break;
  }
#undef HG_K_CASE
#undef HG_K_LAUNCH
  // 33 <= k <= 255: k up to 64 as a compile-time constant
#define HG_KL_LAUNCH(PK, KK)                                                                                                   \
  hipLaunchKernelGGL((kmer_sample_long<KK, PK>), dim3(n_items), dim3(WG), cap_l * sizeof(uint64_t), st, d_seq, d_meta, d_item_genome, \
                     ksize, threshold, seed, canonical ? 1u : 0u, u2t, d_hits, d_cnt, cap_l)
#define HG_KL_CASE(KK)                        \
  case KK:                                    \
    if (packed) HG_KL_LAUNCH(true, KK);       \
    else HG_KL_LAUNCH(false, KK);             \
    return hipGetLastError();
  switch (ksize) {
    HG_KL_CASE(33) HG_KL_CASE(34) HG_KL_CASE(35) HG_KL_CASE(36) HG_KL_CASE(37) HG_KL_CASE(38) HG_KL_CASE(39) HG_KL_CASE(40)
    HG_KL_CASE(41) HG_KL_CASE(42) HG_KL_CASE(43) HG_KL_CASE(44) HG_KL_CASE(45) HG_KL_CASE(46) HG_KL_CASE(47) HG_KL_CASE(48)
    HG_KL_CASE(49) HG_KL_CASE(50) HG_KL_CASE(51) HG_KL_CASE(52) HG_KL_CASE(53) HG_KL_CASE(54) HG_KL_CASE(55) HG_KL_CASE(56)
    HG_KL_CASE(57) HG_KL_CASE(58) HG_KL_CASE(59) HG_KL_CASE(60) HG_KL_CASE(61) HG_KL_CASE(62) HG_KL_CASE(63) HG_KL_CASE(64)
    default:
      break;
  }
#undef HG_KL_CASE
  if (packed) HG_KL_LAUNCH(true, 0);
  else HG_KL_LAUNCH(false, 0);
#undef HG_KL_LAUNCH
  return hipGetLastError();
}
