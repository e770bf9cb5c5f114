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
// HBM-bound byte work: 47 MB in, 82 MB out for 10 000 sketches.
#include <cstring>

#include "hg_internal.h"

namespace {

struct UnpackRow {
  uint64_t off;  // byte offset of the payload (any alignment: it sits behind a path string in the file image)
  uint32_t q, layout;
};

__device__ __forceinline__ uint32_t ld32(const uint8_t *p) {
  return (uint32_t)p[0] | ((uint32_t)p[1] << 8) | ((uint32_t)p[2] << 16) | ((uint32_t)p[3] << 24);
}
__device__ __forceinline__ uint32_t ld16(const uint8_t *p) { return (uint32_t)p[0] | ((uint32_t)p[1] << 8); }

__global__ __launch_bounds__(256) void hg_hv_unpack_kernel(const uint8_t *__restrict__ payloads, const UnpackRow *__restrict__ rows,
                                                           uint32_t hv_d, int16_t *__restrict__ hv) {
  const UnpackRow r = rows[blockIdx.x];
  const uint8_t *__restrict__ src = payloads + r.off;
  int16_t *__restrict__ out = hv + (size_t)blockIdx.x * hv_d;
  const uint32_t q = r.q, mask = (1u << q) - 1u;
  if (r.layout == HG_PAYLOAD_BITPACKER8X) {
    const uint32_t whole = hv_d / 256 * 256;
    const uint16_t offset = (uint16_t)(1u << (q - 1));  // i16 arithmetic: -32768 at q = 16 (src/hd.rs:206)
    for (uint32_t d = threadIdx.x; d < whole; d += 256) {
      const uint32_t i = d & 255u, lane = i & 7u, p = (i >> 3) * q, w0 = p >> 5, c = p & 31u;
      const uint8_t *wp = src + ((size_t)(d >> 8) * 8u * q + 8u * w0 + lane) * 4u;
      uint32_t v = ld32(wp) >> c;
      if (c + q > 32u) v |= ld32(wp + 32) << (32u - c);  // the field straddles two words of its lane (c > 0 here)
      out[d] = (int16_t)(uint16_t)((v & mask) - offset);
    }
    for (uint32_t d = whole + threadIdx.x; d < hv_d; d += 256) out[d] = (int16_t)(uint16_t)(0u - offset);  // src/hd.rs:194
  } else {
    const int16_t half = (int16_t)(uint16_t)(1u << ((q - 1) & 15)), full = (int16_t)(uint16_t)(1u << (q & 15));
    for (uint32_t d = threadIdx.x; d < hv_d; d += 256) {
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

}  // namespace

extern "C" hg_status hg_hv_unpack_batch_dev(hg_ctx *c, const uint8_t *d_payloads, size_t payloads_bytes, const uint64_t *offsets,
                                            const uint8_t *quant_bits, const uint8_t *layouts, size_t n, uint32_t hv_d,
                                            int16_t *d_hv) {
  if (!c) return HG_ERR_INVALID;
  if (n == 0) return HG_OK;
  if (!d_payloads || !offsets || !quant_bits || !d_hv || hv_d == 0) return hg_fail(c, HG_ERR_INVALID, "hg_hv_unpack_batch_dev: bad argument");
  if (n > 0x7FFFFFFFull) return hg_fail(c, HG_ERR_UNSUPPORTED, "more than 2^31 sketches in one call");
  HG_HIP(c, hipSetDevice(c->device));
  hg_status s;
  if ((s = hg_ensure(c, c->w_pktab, n * sizeof(UnpackRow) + 64)) != HG_OK) return s;
  if ((s = hg_ensure_pinned(c, n * sizeof(UnpackRow) + 64)) != HG_OK) return s;
  HG_HIP(c, hipStreamSynchronize(c->stream));  // (the pinned scratch may still feed an earlier upload)
  auto *tab = static_cast<UnpackRow *>(c->h_pin);
  for (size_t i = 0; i < n; ++i) {
    const uint32_t q = quant_bits[i], lay = layouts ? layouts[i] : (uint32_t)HG_PAYLOAD_BITPACKER8X;
    if (q < 1 || q > 16 || lay > (uint32_t)HG_PAYLOAD_NAIVE) return hg_fail(c, HG_ERR_INVALID, "hg_hv_unpack_batch_dev: bad quant_bits / layout");
    const size_t need = lay == (uint32_t)HG_PAYLOAD_NAIVE ? hg_hv_packed_bytes_naive(hv_d, q) : hg_hv_packed_bytes(hv_d, q) / 2 * 2;
    if (offsets[i] > payloads_bytes || need > payloads_bytes - offsets[i])
      return hg_fail(c, HG_ERR_INVALID, "hg_hv_unpack_batch_dev: a payload reaches past the buffer");
    tab[i] = UnpackRow{offsets[i], q, lay};
  }
  HG_HIP(c, hipMemcpyAsync(c->w_pktab.p, tab, n * sizeof(UnpackRow), hipMemcpyHostToDevice, c->stream));
  hipLaunchKernelGGL(hg_hv_unpack_kernel, dim3((unsigned)n), dim3(256), 0, c->stream, d_payloads,
                     static_cast<const UnpackRow *>(c->w_pktab.p), hv_d, d_hv);
  HG_HIP(c, hipGetLastError());
  HG_HIP(c, hipStreamSynchronize(c->stream));  // (the pinned table is free again)
  return HG_OK;
}
