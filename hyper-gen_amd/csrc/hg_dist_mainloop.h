// hg_dist_mainloop.h -- the K loop of dist_mfma_kernel: operand tiles HBM -> LDS (LDS-DMA or register staged), fragments
// LDS -> registers, MFMAs; one barrier per K-step.  Private to hg_dist_kernels.hip.  A variant of the loop is a change to
// THIS file (or a second function beside dist_main_loop): the tile order, the row / column words and the epilogue do not
// see how the accumulators came about.
#pragma once
#include "hg_dist_gemm.h"

namespace {

// Accumulates the tile (row0, col0) of g.A x g.B^T into acc (and, CHUNKED, the finished f32 windows into iacc).
// `publish_tile_words` is called once, behind the first operand stage's loads and in front of the barrier that
// publishes it: the caller stores the tile's row / column words to LDS there (dist_stage_tile_words), so their global
// loads -- issued before this call -- travel with the first stage instead of costing a dependent-load latency later.
template <bool CHUNKED, bool BIG, bool GLDS, int NT, bool I8, bool FP4, class F>
__device__ __forceinline__ void dist_main_loop(const GemmArgs &g, _Float16 *sAB, uint32_t row0, uint32_t col0,
                                               dist_acc_t<I8, FP4> (&acc)[TileCfg<BIG, NT>::WTM][NT],
                                               int32_t (&iacc)[CHUNKED ? TileCfg<BIG, NT>::WTM : 1][CHUNKED ? NT : 1][4],
                                               F &&publish_tile_words) {
  using TC = TileCfg<BIG, NT>;
  constexpr int LROW = GLDS ? BK : LDS_ROW;  // elements per LDS row
  constexpr int BM = TC::BM, BN = TC::BN, WTM = TC::WTM, NWN = TC::NWN, THREADS = TC::THREADS;
  [[maybe_unused]] constexpr int LOADS = TC::LOADS;
  // two LDS stages of (A tile + B tile)
  [[maybe_unused]] constexpr uint32_t A_ELEMS = BM * LROW, B_ELEMS = BN * LROW;
  [[maybe_unused]] constexpr uint32_t TILE_ELEMS = A_ELEMS;           // offset of the B tile inside a stage
  [[maybe_unused]] constexpr uint32_t STAGE_ELEMS = A_ELEMS + B_ELEMS;
  [[maybe_unused]] constexpr uint32_t SROWS = THREADS / 8;            // rows covered by one staging pass
  typedef dist_acc_t<I8, FP4> acc_t;
  static_assert(GLDS || (LOADS == 4 && TC::LOADS_B == 4), "staging macros move 4 pieces per operand");
  static_assert(NT == 4 || GLDS, "wide tiles exist for the LDS-DMA variant only");
  const uint32_t tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const uint32_t wm = wave / NWN, wn = wave % NWN;  // 2 x NWN waves, (WTM*16) x 64 each
  const uint32_t fr = lane & 15, fq = lane >> 4;
  (void)lane;

  // staging: thread t moves 4 x 16 B of A and of B per K-step: row = t/8 + SROWS*i, 16-byte piece t%8
  const uint32_t srow = tid >> 3, spc = tid & 7;
  const _Float16 *gA = g.A + (size_t)(row0 + srow) * g.ldk + spc * 8;
  const _Float16 *gB = g.B + (size_t)(col0 + srow) * g.ldk + spc * 8;
  const size_t rstep = (size_t)SROWS * g.ldk;
  const uint32_t st_off = srow * LROW + spc * 8;                    // this thread's slot in a tile
  // fragment bases in a stage; with the swizzle the lane's 16-byte chunk is (kk*4 + fq) ^ ((row>>1)&7),
  // and (row>>1)&7 == (fr>>1)&7 because all row bases are multiples of 16
  const uint32_t swz = (fr >> 1) & 7;
  const uint32_t fa_off = (wm * WTM * 16 + fr) * LROW + (GLDS ? (fq ^ swz) * 8 : fq * 8);
  const uint32_t fb_off = TILE_ELEMS + (wn * (NT * 16) + fr) * LROW + (GLDS ? (fq ^ swz) * 8 : fq * 8);
  // kk = 1 adds 4 chunks: (4 + fq) ^ swz = (fq ^ swz) ^ 4
  const int32_t kk1_off = GLDS ? ((((fq ^ swz) ^ 4) - (int32_t)(fq ^ swz)) * 8) : 32;
  uint4 ra0, ra1, ra2, ra3, rb0, rb1, rb2, rb3;
#define HG_GLOAD(k0)                                                    \
  ra0 = *reinterpret_cast<const uint4 *>(gA + (k0));                    \
  ra1 = *reinterpret_cast<const uint4 *>(gA + rstep + (k0));            \
  ra2 = *reinterpret_cast<const uint4 *>(gA + 2 * rstep + (k0));        \
  ra3 = *reinterpret_cast<const uint4 *>(gA + 3 * rstep + (k0));        \
  rb0 = *reinterpret_cast<const uint4 *>(gB + (k0));                    \
  rb1 = *reinterpret_cast<const uint4 *>(gB + rstep + (k0));            \
  rb2 = *reinterpret_cast<const uint4 *>(gB + 2 * rstep + (k0));        \
  rb3 = *reinterpret_cast<const uint4 *>(gB + 3 * rstep + (k0));
#define HG_LSTORE(stage)                                                                   \
  {                                                                                        \
    _Float16 *lA = sAB + (stage) * STAGE_ELEMS + st_off, *lB = lA + TILE_ELEMS;            \
    *reinterpret_cast<uint4 *>(lA) = ra0;                                                  \
    *reinterpret_cast<uint4 *>(lA + SROWS * LROW) = ra1;                                \
    *reinterpret_cast<uint4 *>(lA + 2 * SROWS * LROW) = ra2;                            \
    *reinterpret_cast<uint4 *>(lA + 3 * SROWS * LROW) = ra3;                            \
    *reinterpret_cast<uint4 *>(lB) = rb0;                                                  \
    *reinterpret_cast<uint4 *>(lB + SROWS * LROW) = rb1;                                \
    *reinterpret_cast<uint4 *>(lB + 2 * SROWS * LROW) = rb2;                            \
    *reinterpret_cast<uint4 *>(lB + 3 * SROWS * LROW) = rb3;                            \
  }

  // Software pipeline with ONE barrier per K-step.  A step is PHASES phases of 8 MFMAs; the fragments of
  // phase t+1 are read from LDS while phase t multiplies.  The barrier sits BEFORE the last phase of a
  // step, not after it: at that point every fragment of the current stage is already in registers, so
  // once all waves have arrived (and, DMA variant, the next tile has landed: vmcnt(0)) the stage can be
  // refilled and the first fragments of the next stage can be read -- both under the cover of the 8 MFMAs
  // still to issue, instead of an idle matrix pipe right after every barrier.
  //   register-staged: top of step k stores tile k+1 (requested during step k-1) into stage (k+1)&1 and
  //                    requests tile k+2; the barrier in the last phase publishes it.
  //   DMA            : right after the barrier of step k the DMA of tile k+2 starts into stage k&1 (a whole
  //                    step of latency cover).
  const uint32_t nsteps = g.Kp / BK;
  // LDS-DMA staging: thread t fills slots s = i*THREADS + t (i < 4) of each operand tile; slot s is
  // row s/8, LDS chunk s%8, and holds global chunk (s%8) ^ ((row>>1)&7) of that row.  The wave's 64 slots
  // of one instruction are 1 KiB contiguous in LDS, as the DMA requires.
  // The DMA is issued as buffer_load_dwordx4 ... lds through a per-workgroup buffer descriptor (base = the
  // tile's first row, 32-bit per-lane offset, K offset in an SGPR).  The global_load_lds form moves the same
  // bytes, but being FLAT-encoded it makes the compiler flush lgkmcnt to 0 at every LDS dependency while
  // one is in flight, which serialises the fragment reads below with the MFMAs.
  // DMA issue is left to LW of the 8 waves -- one per SIMD when LW = 4: a wave whose VMEM instructions queue
  // up behind the workgroup's burst cannot issue MFMAs meanwhile, and with every wave loading right after the
  // barrier both waves of a SIMD sit in that queue together while the matrix pipe idles.  With one loader
  // per SIMD its partner keeps the pipe busy and the loader catches up while the partner waits at the barrier.
  constexpr int HG_DMA_LOADER_WAVES = 4;
  constexpr int LW = HG_DMA_LOADER_WAVES < THREADS / 64 ? HG_DMA_LOADER_WAVES : THREADS / 64, LT = LW * 64;  // loader waves / threads
  constexpr int PA = BM * 8 / LT, PB = BN * 8 / LT;      // 16-byte pieces per loader thread, A / B tile
  // byte offsets of this thread's pieces inside the A / B row block (fixed-size arrays: a template-sized
  // array here makes hipcc drop the kernel's host stub without a diagnostic)
  uint32_t vA[8], vB[10];
  static_assert(!GLDS || (PA <= 8 && PB <= 10), "piece tables too small");
  __amdgpu_buffer_rsrc_t rsA, rsB;
  if (GLDS) {
    constexpr int PMAX = PA > PB ? PA : PB;
#pragma unroll
    for (int i = 0; i < PMAX; ++i) {
      const uint32_t sl = i * LT + (tid & (LT - 1)), r = sl >> 3, ch = (sl & 7) ^ ((r >> 1) & 7);
      const uint32_t off = (r * g.ldk + ch * 8) * 2;
      if (i < PB) vB[i] = off;
      if (i < PA) vA[i] = off;
    }
    rsA = __builtin_amdgcn_make_buffer_rsrc(const_cast<_Float16 *>(g.A + (size_t)(HG_EXP(512) ? 0u : row0) * g.ldk), 0, 0x7fffffff, 0x00020000);
    rsB = __builtin_amdgcn_make_buffer_rsrc(const_cast<_Float16 *>(g.B + (size_t)(HG_EXP(512) ? 0u : col0) * g.ldk), 0, 0x7fffffff, 0x00020000);
  }
  typedef __attribute__((address_space(3))) void *lds_ptr_t;
#define HG_DMA(stage, k0)                                                                                   \
  {                                                                                                         \
  if (wave < (uint32_t)LW) {                                                                                \
    _Float16 *wbase = sAB + (stage) * STAGE_ELEMS + wave * 64 * 8; /* this wave's 1 KiB of instruction 0 */ \
    _Pragma("unroll") for (int i = 0; i < (PA > PB ? PA : PB); ++i) {                                       \
      if (i < PA)                                                                                           \
        __builtin_amdgcn_raw_ptr_buffer_load_lds(rsA, (lds_ptr_t)(wbase + i * LT * 8), 16, vA[i < PA ? i : 0], (HG_EXP(512) ? (k0) & 1023u : (k0)) * 2, 0, 0); \
      if (i < PB)                                                                                           \
        __builtin_amdgcn_raw_ptr_buffer_load_lds(rsB, (lds_ptr_t)(wbase + TILE_ELEMS + i * LT * 8), 16, vB[i < PB ? i : 0], (HG_EXP(512) ? (k0) & 1023u : (k0)) * 2, 0, 0); \
    }                                                                                                       \
  }                                                                                                         \
  }
  // A fragments per phase: 2 (x NT B fragments = 8..10 MFMAs, the other wave of the SIMD covers the fragment latency)
  constexpr int AF = 2;
  const int32_t fp4_unit_scale = 0x7f7f7f7f;  // FP4: E8M0 block scales of 2^0 for every 32-element block
  constexpr int MP = WTM / AF, PHASES = (BK / 32) * MP;
  half8 bfr[2][NT], afr[2][AF];
  // fragments of phase (kk, mp) of the stage whose fragment bases are pa / pb, into buffer set `buf`
#define HG_FRAGS(buf, pa, pb, kk, mp)                                                                       \
  {                                                                                                         \
    const int32_t ko_ = (kk) ? kk1_off : 0;                                                                 \
    if ((mp) == 0) {                                                                                        \
      _Pragma("unroll") for (int n = 0; n < NT; ++n)                                                         \
          bfr[(kk) & 1][n] = *reinterpret_cast<const half8 *>((pb) + n * 16 * LROW + ko_);                  \
    }                                                                                                       \
    _Pragma("unroll") for (int i_ = 0; i_ < AF; ++i_)                                                       \
        afr[buf][i_] = *reinterpret_cast<const half8 *>((pa) + (AF * (mp) + i_) * 16 * LROW + ko_);         \
  }
  if (GLDS) {
    HG_DMA(0, 0)
  } else {
    HG_GLOAD(0)
    HG_LSTORE(0)
    if (nsteps > 1) {
      HG_GLOAD(BK)
    }
  }
  publish_tile_words();
  __syncthreads();  // (with DMA in flight hipcc's barrier also waits vmcnt(0): stage 0 has landed)
  HG_TSTAMP(1)
  if (GLDS && nsteps > 1) HG_DMA(1, BK)
  if (!HG_EXP(2)) HG_FRAGS(0, sAB + fa_off, sAB + fb_off, 0, 0)
  uint32_t in_chunk = 0;
  for (uint32_t ks = 0; ks < nsteps; ++ks) {
    const uint32_t cur = ks & 1;
    HG_STAMP(0)
    if (!GLDS && ks + 1 < nsteps) {
      HG_LSTORE(cur ^ 1)
      if (ks + 2 < nsteps) {
        const uint32_t k2 = (ks + 2) * BK;
        HG_GLOAD(k2)
      }
    }
    const _Float16 *fA = sAB + cur * STAGE_ELEMS + fa_off, *fB = sAB + cur * STAGE_ELEMS + fb_off;
    const _Float16 *nA = sAB + (cur ^ 1) * STAGE_ELEMS + fa_off, *nB = sAB + (cur ^ 1) * STAGE_ELEMS + fb_off;
#pragma unroll
    for (int t = 0; t < PHASES; ++t) {
      const int kk = t / MP, mp = t % MP;
      if (t + 1 < PHASES) {
        if (!HG_EXP(2) && !(HG_EXP(32) && ks)) HG_FRAGS((t + 1) & 1, fA, fB, (t + 1) / MP, (t + 1) % MP)
      } else {
        // every fragment read of this stage must have returned before another wave may refill it
        HG_STAMP(1)
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        HG_STAMP(2)
        if (!HG_EXP(16)) __syncthreads();
        HG_STAMP(3)
        if (GLDS && ks + 2 < nsteps && !HG_EXP(1)) HG_DMA(cur, (ks + 2) * BK)
        HG_STAMP(4)
        if (ks + 1 < nsteps && !HG_EXP(2) && !HG_EXP(32)) HG_FRAGS(0, nA, nB, 0, 0)
      }
      __builtin_amdgcn_sched_barrier(0);
      if (HG_EXP(8)) {  // fragment reads without the MFMAs
#pragma unroll
        for (int i = 0; i < AF; ++i) asm volatile("" ::"v"(afr[t & 1][i]));
#pragma unroll
        for (int n = 0; n < NT; ++n) asm volatile("" ::"v"(bfr[kk & 1][n]));
      } else if (!HG_EXP(2)) {
#pragma unroll
        for (int i = 0; i < AF; ++i)
#pragma unroll
          for (int n = 0; n < NT; ++n)
            if constexpr (FP4) {  // 32 e2m1 values per lane in the fragment's 16 bytes; block scales 2^0 (E8M0 127).
              // Written as asm: the builtin takes 8-register operand vectors (the e4m3 width); padding the 16-byte
              // fragments to that width costs a copy of every fragment per K-step and 60 VGPRs (spills at 256).
              asm volatile("v_mfma_scale_f32_16x16x128_f8f6f4 %0, %1, %2, %0, %3, %3 op_sel_hi:[0,0,0] cbsz:4 blgp:4"
                           : "+v"(acc[AF * mp + i][n])
                           : "v"(__builtin_bit_cast(int4v, afr[t & 1][i])), "v"(__builtin_bit_cast(int4v, bfr[kk & 1][n])),
                             "v"(fp4_unit_scale));
            } else if constexpr (I8)  // the same 16-byte fragments hold 16 k-consecutive bytes per lane: one instruction covers K = 64
              acc[AF * mp + i][n] = __builtin_amdgcn_mfma_i32_16x16x64_i8(__builtin_bit_cast(int4v, afr[t & 1][i]),
                                                                          __builtin_bit_cast(int4v, bfr[kk & 1][n]),
                                                                          acc[AF * mp + i][n], 0, 0, 0);
            else
              acc[AF * mp + i][n] =
                  __builtin_amdgcn_mfma_f32_16x16x32_f16(afr[t & 1][i], bfr[kk & 1][n], acc[AF * mp + i][n], 0, 0, 0);
      }
      __builtin_amdgcn_sched_barrier(0);
    }
    HG_STAMP(5)
    if (CHUNKED && ++in_chunk == g.chunk_steps) {  // move the exact f32 partial sums into i32
      in_chunk = 0;
#pragma unroll
      for (int m = 0; m < WTM; ++m)
#pragma unroll
        for (int n = 0; n < NT; ++n) {
#pragma unroll
          for (int r = 0; r < 4; ++r) iacc[CHUNKED ? m : 0][CHUNKED ? n : 0][r] += (int32_t)acc[m][n][r];
          acc[m][n] = acc_t{};
        }
    }
  }
  HG_TSTAMP(2)
  if constexpr (FP4)  // asm MFMAs: the hazard recogniser does not know that the accumulators come from the matrix pipe
    asm volatile("s_nop 7\n\ts_nop 7\n\ts_nop 7" ::: "memory");
  // The epilogue reuses the operand stages for its per-wave candidate lists (every fragment read was retired by the
  // last in-loop barrier); phase 2 gathers the norms staged at kernel entry by candidate (from global memory each
  // 64-candidate batch paid a full dependent-load latency: 0.12 ms per launch at 1.3 M hits).
#undef HG_GLOAD
#undef HG_LSTORE
#undef HG_DMA
#undef HG_FRAGS
}

}  // namespace
