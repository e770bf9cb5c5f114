// hg_unpack.hip -- decompress_file_sketch (src/hd.rs:171-232) on the device.
//
// `hyper-gen dist` / `search` start from .sketch files whose hypervectors are bit-packed (4.6 KB per sketch at 9 bits
// against 8 KB of int16).  The reference unpacks them with one rayon task per sketch and keeps the matrices on the host;
// here the file's payload bytes go over the link as they are and one workgroup per sketch decodes them straight into the
// int16 matrix the dist kernels read -- both payload layouts the reference can have written:
//   BitPacker8x (every AVX2 host; src/hd.rs:138-157 / 186-212): blocks of 256 values = 8 lanes x 32 elements, lane l's
//     element r at bit r*q of the lane's stream, stream word w of lane l = the block's u32 number 8 w + l; value =
//     field - 2^(q-1) in i16; dimensions behind the last whole block decode as -2^(q-1);
//   naive (hosts without AVX2; src/hd.rs:158-166 / 213-231): one LSB-first stream of the values' low q bits, decoded
//     with the reference's strict `> 1 << (q-1)` test and its i16 shifts (see include/hypergen.h).
// HBM-bound byte work: 47 MB in, 82 MB out for 10 000 sketches.  A payload sits at ANY byte offset of the file image (behind
// a path string), so the workgroup first copies it into LDS through aligned dword loads, re-aligned on the way
// (v_alignbyte of neighbouring dwords); after that a thread decodes EIGHT consecutive values -- in the BitPacker8x layout
// that is element r of all eight lanes, i.e. eight adjacent stream words (two 16-byte LDS reads, two more if the field
// straddles); in the naive layout q consecutive bytes -- and stores them with one 16-byte write.
#include <algorithm>
#include <cstring>

#include "hg_internal.h"

namespace {

struct UnpackRow {
  uint64_t off;  // byte offset of the payload (any alignment)
  uint32_t q, layout;
};

constexpr uint32_t UNPACK_WG = 256;
// The launch sizes the LDS stage by the largest payload of the call (4.6 KB at hv_d = 4096, q = 9: eight workgroups per CU),
// up to this many bytes (hv_d = 16384 at q = 16, or 32768 at q <= 8); payloads beyond it take the direct path
constexpr uint32_t UNPACK_LDS_MAX = 32768 + 64;

__device__ __forceinline__ uint32_t ld32(const uint8_t *p) {
  return (uint32_t)p[0] | ((uint32_t)p[1] << 8) | ((uint32_t)p[2] << 16) | ((uint32_t)p[3] << 24);
}
__device__ __forceinline__ uint32_t ld16(const uint8_t *p) { return (uint32_t)p[0] | ((uint32_t)p[1] << 8); }

// the direct (unstaged) decoders: one value per thread and trip, byte-wise loads.  Payloads that do not fit the LDS stage.
__device__ void unpack_direct(const uint8_t *__restrict__ src, uint32_t q, uint32_t layout, uint32_t hv_d, int16_t *__restrict__ out) {
  const uint32_t mask = (1u << q) - 1u;
  if (layout == HG_PAYLOAD_BITPACKER8X) {
    const uint32_t whole = hv_d / 256 * 256;
    const uint16_t offset = (uint16_t)(1u << (q - 1));  // i16 arithmetic: -32768 at q = 16 (src/hd.rs:206)
    for (uint32_t d = threadIdx.x; d < whole; d += UNPACK_WG) {
      const uint32_t i = d & 255u, lane = i & 7u, p = (i >> 3) * q, w0 = p >> 5, c = p & 31u;
      const uint8_t *wp = src + ((size_t)(d >> 8) * 8u * q + 8u * w0 + lane) * 4u;
      uint32_t v = ld32(wp) >> c;
      if (c + q > 32u) v |= ld32(wp + 32) << (32u - c);  // the field straddles two words of its lane (c > 0 here)
      out[d] = (int16_t)(uint16_t)((v & mask) - offset);
    }
    for (uint32_t d = whole + threadIdx.x; d < hv_d; d += UNPACK_WG) out[d] = (int16_t)(uint16_t)(0u - offset);  // src/hd.rs:194
  } else {
    const int16_t half = (int16_t)(uint16_t)(1u << ((q - 1) & 15)), full = (int16_t)(uint16_t)(1u << (q & 15));
    for (uint32_t d = threadIdx.x; d < hv_d; d += UNPACK_WG) {
      const uint64_t p = (uint64_t)d * q;
      const uint32_t c = (uint32_t)p & 15u;
      const uint8_t *hp = src + (p >> 4) * 2u;
      uint32_t v = ld16(hp) >> c;
      if (c + q > 16u) v |= ld16(hp + 2) << (16u - c);
      int16_t x = (int16_t)(uint16_t)(v & mask);
      if (x > half) x = (int16_t)((uint16_t)x - (uint16_t)full);  // strictly greater, i16 shifts: src/hd.rs:221-227
      out[d] = x;
    }
  }
}

typedef uint32_t uint4v __attribute__((ext_vector_type(4)));
typedef short short8v __attribute__((ext_vector_type(8)));

__global__ __launch_bounds__(UNPACK_WG) void hg_hv_unpack_kernel(const uint8_t *__restrict__ payloads, const UnpackRow *__restrict__ rows,
                                                                uint32_t hv_d, int16_t *__restrict__ hv, uint32_t lds_bytes) {
  extern __shared__ __attribute__((aligned(16))) uint32_t s_in[];
  const UnpackRow r = rows[blockIdx.x];
  const uint8_t *__restrict__ src = payloads + r.off;
  int16_t *__restrict__ out = hv + (size_t)blockIdx.x * hv_d;
  const uint32_t q = r.q, mask = (1u << q) - 1u, tid = threadIdx.x;
  const bool bp = r.layout == HG_PAYLOAD_BITPACKER8X;
  // bytes the decoders below read: whole BitPacker8x blocks, or the naive stream rounded up to its 16-bit words
  const uint32_t need = bp ? 32u * q * (hv_d / 256) : 2u * (uint32_t)(((uint64_t)q * hv_d + 15) / 16);
  // the 16-byte output stores need the row itself aligned (hv_d a multiple of 8 and the matrix 16-byte aligned)
  const bool staged = need + 32 <= lds_bytes && hv_d % 8 == 0 && ((reinterpret_cast<uintptr_t>(out) & 15) == 0);
  if (!staged) {  // workgroup-uniform
    unpack_direct(src, q, r.layout, hv_d, out);
    return;
  }
  {  // payload -> LDS, re-aligned to a dword boundary: s_in[i] = bytes [4 i, 4 i + 4) of the payload (zero behind its end)
    const uint32_t a = (uint32_t)(reinterpret_cast<uintptr_t>(src) & 3u), n_dw = (need + 3) / 4;
    const uint32_t *__restrict__ g = reinterpret_cast<const uint32_t *>(src - a);  // the aligned dwords around the payload
    const bool head_ok = a == 0 || r.off >= a;  // (the a bytes in front of the payload belong to the caller's buffer)
    for (uint32_t i = tid; i < n_dw; i += UNPACK_WG) {
      // dword i of the payload = the upper 4 - a bytes of aligned dword i and the lower a bytes of aligned dword i + 1;
      // aligned dwords that reach in front of the buffer or behind the payload's last byte are not touched: byte loads there
      const bool fast = (i > 0 || head_ok) && 4u * i + (a ? 8u : 4u) <= need + a;
      uint32_t v;
      if (fast) {
        const uint32_t lo = g[i], hi = a ? g[i + 1] : 0u;
        v = a ? (uint32_t)((((uint64_t)hi << 32) | lo) >> (8u * a)) : lo;
      } else {
        v = 0;
#pragma unroll
        for (uint32_t k = 0; k < 4; ++k)
          if (4u * i + k < need) v |= (uint32_t)src[4u * i + k] << (8u * k);
      }
      s_in[i] = v;
    }
    if (tid < 4) s_in[n_dw + tid] = 0u;  // (readable slack behind the last dword: the naive decoder fetches five dwords per group)
  }
  __syncthreads();
  if (bp) {
    const uint32_t groups = hv_d / 256 * 32;  // 8 consecutive values each = element r of the eight lanes of a block
    const uint16_t offset = (uint16_t)(1u << (q - 1));
    for (uint32_t j = tid; j < groups; j += UNPACK_WG) {
      const uint32_t b = j >> 5, rr = j & 31u, p = rr * q, w0 = p >> 5, c = p & 31u;
      const uint32_t *w = s_in + b * 8u * q + 8u * w0;  // 32-byte aligned
      const uint4v lo0 = *reinterpret_cast<const uint4v *>(w), lo1 = *reinterpret_cast<const uint4v *>(w + 4);
      uint32_t v[8] = {lo0[0] >> c, lo0[1] >> c, lo0[2] >> c, lo0[3] >> c, lo1[0] >> c, lo1[1] >> c, lo1[2] >> c, lo1[3] >> c};
      if (c + q > 32u) {  // (c > 0 here)
        const uint4v hi0 = *reinterpret_cast<const uint4v *>(w + 8), hi1 = *reinterpret_cast<const uint4v *>(w + 12);
        const uint32_t sh = 32u - c;
        v[0] |= hi0[0] << sh, v[1] |= hi0[1] << sh, v[2] |= hi0[2] << sh, v[3] |= hi0[3] << sh;
        v[4] |= hi1[0] << sh, v[5] |= hi1[1] << sh, v[6] |= hi1[2] << sh, v[7] |= hi1[3] << sh;
      }
      short8v o;
#pragma unroll
      for (int k = 0; k < 8; ++k) o[k] = (short)(uint16_t)((v[k] & mask) - offset);
      *reinterpret_cast<short8v *>(out + 8u * j) = o;
    }
    // dimensions behind the last whole block (hv_d % 256 != 0): src/hd.rs:194
    for (uint32_t d = hv_d / 256 * 256 + tid; d < hv_d; d += UNPACK_WG) out[d] = (int16_t)(uint16_t)(0u - offset);
  } else {
    const int16_t half = (int16_t)(uint16_t)(1u << ((q - 1) & 15)), full = (int16_t)(uint16_t)(1u << (q & 15));
    const uint32_t groups = hv_d / 8;  // values [8 j, 8 j + 8) = stream bytes [q j, q j + q)
    for (uint32_t j = tid; j < groups; j += UNPACK_WG) {
      const uint32_t byte0 = j * q, i0 = byte0 >> 2, sh = 8u * (byte0 & 3u);
      // q <= 16 bytes from byte0: five aligned dwords cover them whatever the phase
      const uint32_t d0 = s_in[i0], d1 = s_in[i0 + 1], d2 = s_in[i0 + 2], d3 = s_in[i0 + 3], d4 = s_in[i0 + 4];
      const uint64_t w0 = sh ? (((uint64_t)d1 << 32 | d0) >> sh) | ((uint64_t)d2 << (64 - sh)) : ((uint64_t)d1 << 32 | d0);
      const uint64_t w1 = sh ? (((uint64_t)d3 << 32 | d2) >> sh) | ((uint64_t)d4 << (64 - sh)) : ((uint64_t)d3 << 32 | d2);
      short8v o;
#pragma unroll
      for (int k = 0; k < 8; ++k) {
        const uint32_t bit = (uint32_t)k * q;  // < 128
        uint32_t f;
        if (bit + q <= 64u) f = (uint32_t)(w0 >> bit);
        else if (bit >= 64u) f = (uint32_t)(w1 >> (bit - 64u));
        else f = (uint32_t)(w0 >> bit) | (uint32_t)(w1 << (64u - bit));
        int16_t x = (int16_t)(uint16_t)(f & mask);
        if (x > half) x = (int16_t)((uint16_t)x - (uint16_t)full);  // strictly greater, i16 shifts: src/hd.rs:221-227
        o[k] = x;
      }
      *reinterpret_cast<short8v *>(out + 8u * j) = o;
    }
  }
}

}  // namespace

extern "C" hg_status hg_hv_unpack_batch_dev(hg_ctx *c, const uint8_t *d_payloads, size_t payloads_bytes, const uint64_t *offsets,
                                            const uint8_t *quant_bits, const uint8_t *layouts, size_t n, uint32_t hv_d,
                                            int16_t *d_hv) {
  if (!c) return HG_ERR_INVALID;
  if (n == 0) return HG_OK;
  if (!d_payloads || !offsets || !quant_bits || !d_hv || hv_d == 0) return hg_fail(c, HG_ERR_INVALID, "hg_hv_unpack_batch_dev: bad argument");
  if (n > 0x7FFFFFFFull) return hg_fail(c, HG_ERR_UNSUPPORTED, "more than 2^31 sketches in one call");
  if (hv_d > (1u << 24)) return hg_fail(c, HG_ERR_UNSUPPORTED, "hv_d must be at most 2^24");  // (32-bit byte counts in the kernel)
  HG_ENTER(c);
  hg_status s;
  if ((s = hg_ensure(c, c->w_pktab, n * sizeof(UnpackRow) + 64)) != HG_OK) return s;
  if ((s = hg_ensure_pinned(c, n * sizeof(UnpackRow) + 64)) != HG_OK) return s;
  HG_HIP(c, hipStreamSynchronize(c->stream));  // (the pinned scratch may still feed an earlier upload)
  auto *tab = static_cast<UnpackRow *>(c->h_pin);
  size_t max_need = 0;
  for (size_t i = 0; i < n; ++i) {
    const uint32_t q = quant_bits[i], lay = layouts ? layouts[i] : (uint32_t)HG_PAYLOAD_BITPACKER8X;
    if (q < 1 || q > 16 || lay > (uint32_t)HG_PAYLOAD_NAIVE) return hg_fail(c, HG_ERR_INVALID, "hg_hv_unpack_batch_dev: bad quant_bits / layout");
    const size_t need = lay == (uint32_t)HG_PAYLOAD_NAIVE ? hg_hv_packed_bytes_naive(hv_d, q) : hg_hv_packed_bytes(hv_d, q) / 2 * 2;
    if (offsets[i] > payloads_bytes || need > payloads_bytes - offsets[i])
      return hg_fail(c, HG_ERR_INVALID, "hg_hv_unpack_batch_dev: a payload reaches past the buffer");
    tab[i] = UnpackRow{offsets[i], q, lay};
    max_need = std::max(max_need, need);
  }
  const uint32_t lds_bytes = (uint32_t)std::min<size_t>((max_need + 32 + 255) & ~(size_t)255, UNPACK_LDS_MAX);
  HG_HIP(c, hipMemcpyAsync(c->w_pktab.p, tab, n * sizeof(UnpackRow), hipMemcpyHostToDevice, c->stream));
  hipLaunchKernelGGL(hg_hv_unpack_kernel, dim3((unsigned)n), dim3(UNPACK_WG), lds_bytes, c->stream, d_payloads,
                     static_cast<const UnpackRow *>(c->w_pktab.p), hv_d, d_hv, lds_bytes);
  HG_HIP(c, hipGetLastError());
  HG_HIP(c, hipStreamSynchronize(c->stream));  // (the pinned table is free again)
  return HG_OK;
}
