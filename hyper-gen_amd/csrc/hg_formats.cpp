// hg_formats.cpp -- host-side data formats either side of the device path:
//   * lossless HV bit-packing   (hd::compress_hd_sketch / decompress_hd_sketch, src/hd.rs:114-232)
//   * the .sketch container     (bincode 1.3 of Vec<FileSketch>, src/types.rs:224-235,
//                                src/utils.rs:234-258)
//   * FASTA -> merged sequence  (fastx_reader::read_merge_seq, src/fastx_reader.rs:6-29)
// The BitPacker8x and bincode layouts are restated from the published crate algorithms; no
// reference-produced .sketch file exists in this environment, so byte compatibility with the
// Rust binary is UNPINNED ("parity status").
#include <fcntl.h>
#include <sched.h>
#include <sys/syscall.h>
#include <sys/stat.h>
#include <unistd.h>
#include <zlib.h>
#if defined(__x86_64__)
#include <immintrin.h>
#endif

#include <algorithm>
#include <cerrno>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <new>
#include <string>
#include <vector>

#include "../../include/hypergen.h"
#include "hg_host.h"

extern "C" uint32_t hg_hv_quant_bits(const int16_t *hv, uint32_t hv_d) {
  if (!hv || hv_d == 0) return 6;
  int16_t mn = hv[0], mx = hv[0];
  for (uint32_t d = 1; d < hv_d; ++d) {
    mn = hv[d] < mn ? hv[d] : mn;
    mx = hv[d] > mx ? hv[d] : mx;
  }
  uint32_t q = 6;  // src/hd.rs:123-136: widen until [-2^(q-1), 2^(q-1)-1] covers [min, max]
  while (q < 16) {
    const int32_t lo = -(1 << (q - 1)), hi = (1 << (q - 1)) - 1;
    if (lo <= mn && hi >= mx) break;
    ++q;
  }
  return q;
}

namespace {
// BitPacker8x block: 256 values = 32 rows x 8 lanes; lane l packs its 32 values LSB-first into
// q u32 words, word w of lane l is output u32 number 8*w + l; values are OR-ed in unmasked.
void pack_block(const uint32_t *in, uint32_t q, uint32_t *out) {
  for (uint32_t l = 0; l < 8; ++l) {
    uint32_t acc = 0, w = 0;
    for (uint32_t r = 0; r < 32; ++r) {
      const uint32_t v = in[8 * r + l], cur = (r * q) & 31;
      acc = cur ? (acc | (v << cur)) : v;
      const uint32_t remaining = 32 - cur;
      if (remaining <= q) {
        out[8 * w + l] = acc;
        ++w;
        acc = remaining < q ? (v >> remaining) : 0;
      }
    }
  }
}
void unpack_block(const uint32_t *in, uint32_t q, uint32_t *out) {
  const uint32_t mask = q >= 32 ? 0xffffffffu : ((1u << q) - 1);
  for (uint32_t l = 0; l < 8; ++l)
    for (uint32_t r = 0; r < 32; ++r) {
      const uint32_t bit = r * q, w = bit >> 5, cur = bit & 31;
      uint32_t v = in[8 * w + l] >> cur;
      if (cur + q > 32) v |= in[8 * (w + 1) + l] << (32 - cur);
      out[8 * r + l] = v & mask;
    }
}
}  // namespace

extern "C" size_t hg_hv_packed_bytes(uint32_t hv_d, uint32_t q) { return (size_t)q * (hv_d >> 3); }  // src/hd.rs:146

extern "C" hg_status hg_hv_pack(const int16_t *hv, uint32_t hv_d, uint32_t q, uint8_t *packed) {
  if (!hv || !packed || q < 1 || q > 16) return HG_ERR_INVALID;
  const int16_t offset = (int16_t)(1 << (q - 1));  // i16 arithmetic, src/hd.rs:140-141
  uint32_t blk[256], out[8 * 16];
  // Only whole 256-blocks are packed (src/hd.rs:147 `hv_d / BLOCK_LEN`); the reference leaves the rest of its
  // zero-initialised `quant_bit * (hv_d >> 3)` bytes untouched, so the dimensions behind the last block are LOST
  const size_t tail0 = (size_t)32 * q * (hv_d / 256), total = hg_hv_packed_bytes(hv_d, q);
  if (total > tail0) std::memset(packed + tail0, 0, total - tail0);
  for (uint32_t b = 0; b < hv_d / 256; ++b) {
    for (uint32_t i = 0; i < 256; ++i) blk[i] = (uint32_t)(int32_t)(int16_t)(hv[b * 256 + i] + offset);
    pack_block(blk, q, out);
    std::memcpy(packed + (size_t)32 * q * b, out, (size_t)32 * q);
  }
  return HG_OK;
}

extern "C" hg_status hg_hv_unpack(const uint8_t *packed, uint32_t hv_d, uint32_t q, int16_t *hv) {
  if (!hv || !packed || q < 1 || q > 16) return HG_ERR_INVALID;
  // dimensions behind the last whole block decode from the reference's zero-initialised u32 vector
  // (src/hd.rs:194,206-212): 0 as i16 - offset
  for (uint32_t d = hv_d / 256 * 256; d < hv_d; ++d) hv[d] = (int16_t)(0 - (int16_t)(1 << (q - 1)));
  const int16_t offset = (int16_t)(1 << (q - 1));
  uint32_t blk[256], in[8 * 16];
  for (uint32_t b = 0; b < hv_d / 256; ++b) {
    std::memcpy(in, packed + (size_t)32 * q * b, (size_t)32 * q);
    unpack_block(in, q, blk);
    for (uint32_t i = 0; i < 256; ++i) hv[b * 256 + i] = (int16_t)((int16_t)blk[i] - offset);  // :206-212
  }
  return HG_OK;
}

// ---- the non-AVX2 layout (src/hd.rs:158-166, 213-231) ------------------------------------------------------------------
// A plain LSB-first bit stream of the values' low q bits in 16-bit words, (q*hv_d + 16) / 16 of them.  Written here
// through a 64-bit shift register, a value at a time (the oracle follows the reference's bit-by-bit loops).
extern "C" size_t hg_hv_packed_bytes_naive(uint32_t hv_d, uint32_t q) { return 2 * (((size_t)q * hv_d + 16) / 16); }

extern "C" hg_status hg_hv_pack_naive(const int16_t *hv, uint32_t hv_d, uint32_t q, uint8_t *packed) {
  if (!hv || !packed || q < 1 || q > 16) return HG_ERR_INVALID;
  const size_t words = hg_hv_packed_bytes_naive(hv_d, q) / 2;
  const uint32_t mask = (1u << q) - 1u;
  uint64_t reg = 0;
  uint32_t have = 0;
  size_t w = 0;
  auto put16 = [&](uint16_t v) {
    packed[2 * w] = (uint8_t)v, packed[2 * w + 1] = (uint8_t)(v >> 8);  // i16 little endian, as bincode stores them
    ++w;
  };
  for (uint32_t d = 0; d < hv_d; ++d) {
    reg |= (uint64_t)((uint32_t)(uint16_t)hv[d] & mask) << have;
    for (have += q; have >= 16; have -= 16, reg >>= 16) put16((uint16_t)reg);
  }
  if (have) put16((uint16_t)reg);
  while (w < words) put16(0);
  return HG_OK;
}

extern "C" hg_status hg_hv_unpack_naive(const uint8_t *packed, uint32_t hv_d, uint32_t q, int16_t *hv) {
  if (!hv || !packed || q < 1 || q > 16) return HG_ERR_INVALID;
  const uint32_t mask = (1u << q) - 1u;
  // `1 << (q-1)` and `1 << q` as the reference's i16 expressions evaluate in a release build (shift amounts modulo 16)
  const int16_t half = (int16_t)(uint16_t)(1u << ((q - 1) & 15)), full = (int16_t)(uint16_t)(1u << (q & 15));
  uint64_t reg = 0;
  uint32_t have = 0;
  size_t w = 0;
  for (uint32_t d = 0; d < hv_d; ++d) {
    for (; have < q; have += 16, ++w) reg |= (uint64_t)((uint32_t)packed[2 * w] | ((uint32_t)packed[2 * w + 1] << 8)) << have;
    int16_t v = (int16_t)(uint16_t)((uint32_t)reg & mask);
    reg >>= q, have -= q;
    if (v > half) v = (int16_t)((uint16_t)v - (uint16_t)full);  // strictly greater: src/hd.rs:221-227
    hv[d] = v;
  }
  return HG_OK;
}

extern "C" int hg_hv_payload_layout(uint32_t hv_d, uint32_t q, size_t payload_bytes) {
  if (q < 1 || q > 16) return -1;
  // (the i16 view of the BitPacker8x bytes drops an odd last byte: src/hd.rs:155-157 `align_to::<i16>().1`)
  if (payload_bytes == hg_hv_packed_bytes(hv_d, q) / 2 * 2) return HG_PAYLOAD_BITPACKER8X;
  if (payload_bytes == hg_hv_packed_bytes_naive(hv_d, q)) return HG_PAYLOAD_NAIVE;
  return -1;
}

// ---- .sketch: bincode 1.x default config = little endian, fixed-width ints, u64 lengths --------------
struct hg_sketch_file {
  std::vector<hg_file_sketch> recs;
  std::vector<std::string> names;
  std::vector<std::vector<int16_t>> payloads;  // hg_sketch_file_read
  std::vector<uint8_t> image;                  // hg_sketch_file_read_image: the file as read ...
  std::vector<uint64_t> payload_off;           // ... and where record i's payload starts in it
};

namespace {
template <class T>
void put(std::vector<uint8_t> &o, T v) {
  uint8_t b[sizeof(T)];
  std::memcpy(b, &v, sizeof(T));
  o.insert(o.end(), b, b + sizeof(T));
}
template <class T>
bool get(const uint8_t *&p, const uint8_t *end, T &v) {
  if ((size_t)(end - p) < sizeof(T)) return false;
  std::memcpy(&v, p, sizeof(T));
  p += sizeof(T);
  return true;
}
}  // namespace

extern "C" hg_status hg_sketch_file_write(const char *path, const hg_file_sketch *recs, size_t n) {
  if (!path || (n && !recs)) return HG_ERR_INVALID;
  std::vector<uint8_t> o;
  put<uint64_t>(o, n);  // Vec length
  for (size_t i = 0; i < n; ++i) {
    const hg_file_sketch &r = recs[i];  // field order of src/types.rs:224-235
    put<uint8_t>(o, r.ksize);
    put<uint64_t>(o, r.scaled);
    put<uint8_t>(o, r.canonical ? 1 : 0);
    put<uint64_t>(o, r.seed);
    put<uint64_t>(o, r.hv_d);  // usize
    put<uint8_t>(o, r.hv_quant_bits);
    put<int32_t>(o, r.hv_norm_2);
    const size_t sl = r.file_str ? std::strlen(r.file_str) : 0;
    put<uint64_t>(o, sl);
    o.insert(o.end(), (const uint8_t *)r.file_str, (const uint8_t *)r.file_str + sl);
    put<uint64_t>(o, r.hv_len);
    const uint8_t *hb = reinterpret_cast<const uint8_t *>(r.hv);
    o.insert(o.end(), hb, hb + r.hv_len * 2);
  }
  FILE *f = std::fopen(path, "wb");
  if (!f) return HG_ERR_IO;
  const bool ok = o.empty() || std::fwrite(o.data(), 1, o.size(), f) == o.size();
  return (std::fclose(f) == 0 && ok) ? HG_OK : HG_ERR_IO;
}

namespace {
// the whole file in one buffer (fstat + read: no 64 KiB fread loop, no vector growth)
hg_status slurp(const char *path, std::vector<uint8_t> &buf) {
  const int fd = ::open(path, O_RDONLY | O_CLOEXEC);
  if (fd < 0) return HG_ERR_IO;
  struct stat st;
  if (::fstat(fd, &st) != 0 || st.st_size < 0) {
    ::close(fd);
    return HG_ERR_IO;
  }
  try {
    buf.resize((size_t)st.st_size);
  } catch (const std::bad_alloc &) {
    ::close(fd);
    return HG_ERR_OOM;
  }
  size_t have = 0;
  while (have < buf.size()) {
    const ssize_t got = ::read(fd, buf.data() + have, buf.size() - have);
    if (got < 0 && errno == EINTR) continue;
    if (got <= 0) break;
    have += (size_t)got;
  }
  ::close(fd);
  buf.resize(have);
  return HG_OK;
}

// copy_payloads = false: the records' hv stay NULL, the payloads are addressed through image + payload_off
// (hg_sketch_file_image / hg_sketch_file_payload_offset: what goes to hg_hv_unpack_batch_dev)
hg_status read_sketch_file(const char *path, hg_sketch_file **out, bool copy_payloads) {
  if (!path || !out) return HG_ERR_INVALID;
  *out = nullptr;
  hg_sketch_file *sf = new (std::nothrow) hg_sketch_file();
  if (!sf) return HG_ERR_OOM;
  hg_status rs = slurp(path, sf->image);
  if (rs != HG_OK) {
    delete sf;
    return rs;
  }
  const std::vector<uint8_t> &buf = sf->image;
  const uint8_t *p = buf.data(), *end = p + buf.size();
  uint64_t n = 0;
  // a record is at least 47 bytes (31 fixed + two u64 lengths): a count the file cannot hold is corruption,
  // not an allocation request (a hostile count must not reach resize())
  if (!get(p, end, n) || n > (buf.size() - 8) / 47) {
    delete sf;
    return HG_ERR_IO;
  }
  try {
  sf->recs.resize(n), sf->names.resize(n), sf->payload_off.resize(n);
  if (copy_payloads) sf->payloads.resize(n);
  for (uint64_t i = 0; i < n; ++i) {
    hg_file_sketch &r = sf->recs[i];
    std::memset(&r, 0, sizeof r);
    uint64_t sl = 0, hl = 0;
    bool ok = get(p, end, r.ksize) && get(p, end, r.scaled) && get(p, end, r.canonical) &&
              get(p, end, r.seed) && get(p, end, r.hv_d) && get(p, end, r.hv_quant_bits) &&
              get(p, end, r.hv_norm_2) && get(p, end, sl) && sl <= (uint64_t)(end - p);
    if (ok) {
      sf->names[i].assign(reinterpret_cast<const char *>(p), sl);
      p += sl;
      ok = get(p, end, hl) && hl <= (uint64_t)(end - p) / 2;
    }
    if (!ok) {
      delete sf;
      return HG_ERR_IO;
    }
    sf->payload_off[i] = (uint64_t)(p - buf.data());
    if (copy_payloads) {
      sf->payloads[i].resize(hl);
      if (hl) std::memcpy(sf->payloads[i].data(), p, hl * 2);
    }
    p += hl * 2;
    r.hv_len = hl;
  }
  } catch (const std::bad_alloc &) {  // nothing may unwind through the C ABI
    delete sf;
    return HG_ERR_OOM;
  }
  for (uint64_t i = 0; i < n; ++i) {  // pointers only after the vectors stopped moving
    sf->recs[i].file_str = sf->names[i].c_str();
    sf->recs[i].hv = copy_payloads ? sf->payloads[i].data() : nullptr;
  }
  if (copy_payloads) std::vector<uint8_t>().swap(sf->image);  // (the copies are the payloads now)
  *out = sf;
  return HG_OK;
}
}  // namespace

extern "C" hg_status hg_sketch_file_read(const char *path, hg_sketch_file **out) { return read_sketch_file(path, out, true); }
extern "C" hg_status hg_sketch_file_read_image(const char *path, hg_sketch_file **out) { return read_sketch_file(path, out, false); }
extern "C" const uint8_t *hg_sketch_file_image(const hg_sketch_file *f, size_t *bytes) {
  if (bytes) *bytes = f ? f->image.size() : 0;
  return (f && !f->image.empty()) ? f->image.data() : nullptr;
}
extern "C" uint64_t hg_sketch_file_payload_offset(const hg_sketch_file *f, size_t i) {
  return (f && i < f->payload_off.size()) ? f->payload_off[i] : 0;
}

extern "C" size_t hg_sketch_file_count(const hg_sketch_file *f) { return f ? f->recs.size() : 0; }
extern "C" const hg_file_sketch *hg_sketch_file_get(const hg_sketch_file *f, size_t i) {
  return (f && i < f->recs.size()) ? &f->recs[i] : nullptr;
}
extern "C" void hg_sketch_file_free(hg_sketch_file *f) { delete f; }

// ---- 2-bit packing of sequence for the PCIe link ------------------------------------------------------------
// Blob layout for n bases: codes, 4 bases per byte (base i in bits 2(i&3)..2(i&3)+1 of byte i>>2; A,C,G,T = 0..3,
// the device's own code ((x>>1)^(x>>2))&3), padded to a multiple of 16 bytes; then the NOT-a-base mask, 1 bit per
// base (bit i&7 of byte i>>3), padded likewise.  3 bits per base = 0.375 of the ASCII bytes; hg_unpack2_dev turns
// it back into the ASCII the k-mer kernels classify the same way ('A','C','G','T' / 'N'), so results cannot differ.
namespace {
inline size_t al16(size_t x) { return (x + 15) & ~(size_t)15; }

inline void pack2_scalar(const uint8_t *seq, size_t i0, size_t i1, bool u2t, uint8_t *codes, uint8_t *mask) {
  for (size_t i = i0; i < i1; ++i) {  // i0 is a multiple of 8
    const uint8_t x = seq[i], u = x & 0xDF;
    const bool ok = u == 'A' || u == 'C' || u == 'G' || u == 'T' || (u2t && u == 'U');
    const uint8_t c = ok ? (uint8_t)(((x >> 1) ^ (x >> 2)) & 3) : 0;
    if ((i & 3) == 0) codes[i >> 2] = 0;
    if ((i & 7) == 0) mask[i >> 3] = 0;
    codes[i >> 2] |= (uint8_t)(c << (2 * (i & 3)));
    mask[i >> 3] |= (uint8_t)((ok ? 0 : 1) << (i & 7));
  }
}

#if defined(__x86_64__)
__attribute__((target("avx2"))) size_t pack2_avx2(const uint8_t *seq, size_t n, bool u2t, uint8_t *codes, uint8_t *mask) {
  const __m256i up = _mm256_set1_epi8((char)0xDF), three = _mm256_set1_epi8(3);
  const __m256i cA = _mm256_set1_epi8('A'), cC = _mm256_set1_epi8('C'), cG = _mm256_set1_epi8('G'), cT = _mm256_set1_epi8('T');
  const __m256i cU = _mm256_set1_epi8(u2t ? 'U' : 'T');
  const __m256i w14 = _mm256_set1_epi16(0x0401), w116 = _mm256_set1_epi32(0x00100001);
  const __m256i pick = _mm256_setr_epi8(0, 4, 8, 12, -1, -1, -1, -1, -1, -1, -1, -1, -1, -1, -1, -1,
                                        0, 4, 8, 12, -1, -1, -1, -1, -1, -1, -1, -1, -1, -1, -1, -1);
  size_t i = 0;
  for (; i + 32 <= n; i += 32) {
    const __m256i x = _mm256_loadu_si256(reinterpret_cast<const __m256i *>(seq + i));
    const __m256i u = _mm256_and_si256(x, up);
    const __m256i ok = _mm256_or_si256(_mm256_or_si256(_mm256_cmpeq_epi8(u, cA), _mm256_cmpeq_epi8(u, cC)),
                                       _mm256_or_si256(_mm256_or_si256(_mm256_cmpeq_epi8(u, cG), _mm256_cmpeq_epi8(u, cT)),
                                                       _mm256_cmpeq_epi8(u, cU)));
    __m256i c = _mm256_and_si256(_mm256_xor_si256(_mm256_srli_epi16(x, 1), _mm256_srli_epi16(x, 2)), three);
    c = _mm256_and_si256(c, ok);  // non-bases carry code 0
    const __m256i q = _mm256_madd_epi16(_mm256_maddubs_epi16(c, w14), w116);  // c0 + 4 c1 + 16 c2 + 64 c3 per dword
    const __m256i b = _mm256_shuffle_epi8(q, pick);
    const uint32_t lo = (uint32_t)_mm256_cvtsi256_si32(b), hi = (uint32_t)_mm256_extract_epi32(b, 4);
    const uint32_t bad = ~(uint32_t)_mm256_movemask_epi8(ok);
    // the stores stay behind the loads also when codes == seq (8 + 4 bytes written per 32 read)
    std::memcpy(codes + (i >> 2), &lo, 4), std::memcpy(codes + (i >> 2) + 4, &hi, 4);
    std::memcpy(mask + (i >> 3), &bad, 4);
  }
  return i;
}

// AVX-512 (BW + VBMI) form of the same: one 128-entry table look-up classifies and encodes 64 bases (entry = the 2-bit code,
// or 0x80 for a byte that is not a base; bytes >= 0x80 carry their own sign bit), two multiply-adds gather 4 codes per
// dword, vpmovdb narrows them to 16 code bytes; the sign bits are the 64 mask bits.  ~2.5x the AVX2 loop per core.
#ifndef HG_PACK_NT
#define HG_PACK_NT 1  /* 0: ordinary stores also for streamed output (A/B) */
#endif
// STREAM: the output is not read by this core again (it goes to the device by DMA) -- non-temporal stores, so that the
// destination lines are not fetched from memory before they are overwritten (codes 16-byte, mask 8-byte aligned; the
// caller fences).  16 threads packing a host-fed batch are bound by host memory traffic, of which that fetch was a sixth.
template <bool STREAM>
__attribute__((target("avx512f,avx512bw,avx512vbmi"))) size_t pack2_avx512(const uint8_t *seq, size_t n, bool u2t, uint8_t *codes,
                                                                          uint8_t *mask) {
  alignas(64) uint8_t lut[128];
  std::memset(lut, 0x80, sizeof lut);
  for (const char *b = u2t ? "ACGTU" : "ACGT"; *b; ++b) {
    const uint8_t x = (uint8_t)*b, code = (uint8_t)(((x >> 1) ^ (x >> 2)) & 3);
    lut[x] = code, lut[x | 0x20] = code;
  }
  const __m512i lo = _mm512_load_si512(lut), hi = _mm512_load_si512(lut + 64);
  const __m512i w14 = _mm512_set1_epi16(0x0401), w116 = _mm512_set1_epi32(0x00100001);
  size_t i = 0;
  for (; i + 64 <= n; i += 64) {
    const __m512i x = _mm512_loadu_si512(seq + i);
    const __m512i t = _mm512_permutex2var_epi8(lo, x, hi);                   // index = low 7 bits of x
    const __mmask64 bad = _mm512_movepi8_mask(_mm512_or_si512(t, x));       // not a base, or a byte >= 0x80
    const __m512i c = _mm512_maskz_mov_epi8(~bad, t);                       // non-bases carry code 0
    const __m512i q = _mm512_madd_epi16(_mm512_maddubs_epi16(c, w14), w116);  // c0 + 4 c1 + 16 c2 + 64 c3 per dword
    const __m128i b = _mm512_cvtepi32_epi8(q);
    const uint64_t m = (uint64_t)bad;
    // the stores stay behind the loads also when codes == seq (16 + 8 bytes written per 64 read)
    if (STREAM && HG_PACK_NT) {
      _mm_stream_si128(reinterpret_cast<__m128i *>(codes + (i >> 2)), b);
      _mm_stream_si64(reinterpret_cast<long long *>(mask + (i >> 3)), (long long)m);
    } else {
      _mm_storeu_si128(reinterpret_cast<__m128i *>(codes + (i >> 2)), b);
      std::memcpy(mask + (i >> 3), &m, 8);
    }
  }
  return i;
}
#endif
}  // namespace

extern "C" size_t hg_pack2_size(size_t n_bps) { return al16((n_bps + 3) / 4) + al16((n_bps + 7) / 8); }

namespace {
// bases [0, n) of `seq` -> codes / mask, both pointing at the bytes of base 0 (n a multiple of 32, or the tail)
// (stream: see pack2_avx512 -- only when codes is 16-byte and mask 8-byte aligned and neither overlaps seq)
inline void pack2_span(const uint8_t *seq, size_t n, bool u2t, uint8_t *codes, uint8_t *mask, bool stream = false) {
  size_t done = 0;
#if defined(__x86_64__)
#ifndef HG_PACK_AVX512
#define HG_PACK_AVX512 1  /* 0: the AVX2 loop also where AVX-512 VBMI exists (A/B) */
#endif
  static const int isa = (HG_PACK_AVX512 && __builtin_cpu_supports("avx512bw") && __builtin_cpu_supports("avx512vbmi")) ? 2
                         : __builtin_cpu_supports("avx2") ? 1 : 0;
  if (isa == 2) {
    if (stream && !((uintptr_t)codes & 15) && !((uintptr_t)mask & 7)) {
      done = pack2_avx512<true>(seq, n, u2t, codes, mask);
      _mm_sfence();
    } else {
      done = pack2_avx512<false>(seq, n, u2t, codes, mask);
    }
  }
  if (isa >= 1) done += pack2_avx2(seq + done, n - done, u2t, codes + (done >> 2), mask + (done >> 3));
#endif
  pack2_scalar(seq, done, n, u2t, codes, mask);
}

// the not-a-base bits are collected aside (so that packing in place works) and appended once the length is known;
// kept per thread: a fresh 0.6 MB block per 5 Mbp genome is an mmap / page-fault / munmap cycle each time, and
// reader threads then queue on the address-space lock
uint8_t *thread_mask(size_t bytes) {
  static thread_local std::vector<uint8_t> mask;
  try {
    if (mask.size() < bytes + 64) mask.resize(bytes + bytes / 4 + 64);
  } catch (const std::bad_alloc &) {
    return nullptr;
  }
  return mask.data();
}

// zero padding of the code area, then the mask behind it
void pack2_finish(uint8_t *out, size_t n_bps, uint8_t *mask) {
  const size_t cb = al16((n_bps + 3) / 4), mb = al16((n_bps + 7) / 8), used = (n_bps + 3) / 4, mused = (n_bps + 7) / 8;
  if (cb > used) std::memset(out + used, 0, cb - used);
  if (mb > mused) std::memset(mask + mused, 0, mb - mused);
  std::memcpy(out + cb, mask, mb);
}
}  // namespace

// Sparse form for the PCIe link: the same codes, but the not-a-base positions as a sorted table of runs instead of a bit
// per base -- an assembly has a handful of them (N gaps between contigs, the 'N' per record start), so the link carries
// 0.25 bytes per base instead of 0.375.  Layout: [codes, padded to 16 bytes][u32 n_runs, u32 0, n_runs x {u32 start,
// u32 length}, padded to 16 bytes].  The device rebuilds the bitmap from the table (hg_stream.hip: expand_runs_kernel).
extern "C" size_t hg_pack2s_size(size_t n_bps, size_t n_runs) { return al16((n_bps + 3) / 4) + al16(8 + 8 * n_runs); }

extern "C" hg_status hg_pack2s(const uint8_t *seq, size_t n_bps, uint32_t norm_mode, uint8_t *out, size_t cap, size_t *size_out) {
  if ((n_bps && !seq) || !out || !size_out || norm_mode > HG_NORM_U2T) return HG_ERR_INVALID;
  if (n_bps >= ((size_t)1 << 32)) return HG_ERR_UNSUPPORTED;  // (32-bit run positions; such a genome takes hg_pack2)
  const size_t cb = al16((n_bps + 3) / 4), used = (n_bps + 3) / 4, mwords = (n_bps + 63) / 64;
  *size_out = cb + 16;
  if (cap < cb + 16) return HG_ERR_CAPACITY;
  uint8_t *mask = thread_mask(8 * mwords + 64);
  if (!mask) return HG_ERR_OOM;
  pack2_span(seq, n_bps, norm_mode == HG_NORM_U2T, out, mask);
  if (cb > used) std::memset(out + used, 0, cb - used);
  if (n_bps & 63) std::memset(mask + (n_bps + 7) / 8, 0, 8 * mwords - (n_bps + 7) / 8);  // whole 64-bit words below
  // runs of set bits, in order (bits at or behind n_bps are zero: pack2_span sets none, the tail was cleared above)
  const size_t max_runs = (cap - cb - 8) / 8;
  uint32_t *tab = reinterpret_cast<uint32_t *>(out + cb);
  size_t n_runs = 0;
  bool open = false;
  uint64_t start = 0;
  auto emit = [&](uint64_t st, uint64_t len) {
    if (n_runs < max_runs) tab[2 + 2 * n_runs] = (uint32_t)st, tab[3 + 2 * n_runs] = (uint32_t)len;
    ++n_runs;
  };
  for (size_t w = 0; w < mwords; ++w) {
    uint64_t x;
    std::memcpy(&x, mask + 8 * w, 8);
    if (open) {  // a run came in from the word below: it ends at this word's first zero bit
      if (x == ~(uint64_t)0) continue;
      const unsigned z = (unsigned)__builtin_ctzll(~x);
      emit(start, 64 * w + z - start);
      open = false;
      x = z ? (x >> z) << z : x;
    }
    while (x) {
      const unsigned p = (unsigned)__builtin_ctzll(x);
      const uint64_t y = ~(x >> p);  // (the shift fills zeros from the top: a run that reaches bit 63 gives 64 - p)
      const unsigned len = y ? (unsigned)__builtin_ctzll(y) : 64u;
      if (p + len >= 64) {
        start = 64 * w + p, open = true;
        break;
      }
      emit(64 * w + p, len);
      x &= ~((((uint64_t)1 << len) - 1) << p);
    }
  }
  if (open) {  // a run that ends with the sequence
    emit(start, n_bps - start);
  }
  *size_out = hg_pack2s_size(n_bps, n_runs);
  if (n_runs > max_runs || *size_out > cap) return HG_ERR_CAPACITY;
  tab[0] = (uint32_t)n_runs, tab[1] = 0;
  const size_t tb = 8 + 8 * n_runs;
  if (al16(tb) > tb) std::memset(out + cb + tb, 0, al16(tb) - tb);
  return HG_OK;
}

// `out` may be `seq` itself (packing in place): the codes trail the reads, the mask is collected aside.
extern "C" hg_status hg_pack2(const uint8_t *seq, size_t n_bps, uint32_t norm_mode, uint8_t *out) {
  if ((n_bps && !seq) || !out || norm_mode > HG_NORM_U2T) return HG_ERR_INVALID;
  uint8_t *mask = thread_mask(al16((n_bps + 7) / 8));
  if (!mask) return HG_ERR_OOM;
  pack2_span(seq, n_bps, norm_mode == HG_NORM_U2T, out, mask);
  pack2_finish(out, n_bps, mask);
  return HG_OK;
}

// Bases [b0, b1) of a genome of n_bps bases into their place in the genome's blob at `out` (b0 a multiple of 64, b1 a
// multiple of 64 or n_bps; the piece that ends the genome also writes the two paddings): several host threads pack one
// genome, or many, in pieces of even size.  `out` must not overlap `seq`.
void hg_pack2_piece(const uint8_t *seq, size_t n_bps, uint32_t norm_mode, uint8_t *out, size_t b0, size_t b1) {
  const size_t cb = al16((n_bps + 3) / 4);
  uint8_t *mask = out + cb;
  pack2_span(seq + b0, b1 - b0, norm_mode == HG_NORM_U2T, out + (b0 >> 2), mask + (b0 >> 3), true);
  if (b1 == n_bps) {
    const size_t mb = al16((n_bps + 7) / 8), used = (n_bps + 3) / 4, mused = (n_bps + 7) / 8;
    if (cb > used) std::memset(out + used, 0, cb - used);
    if (mb > mused) std::memset(mask + mused, 0, mb - mused);
  }
}

// ---- NUMA placement of host threads ----------------------------------------------------------------------------------
extern "C" int hg_bind_thread_to_numa_node(int node, unsigned threads_sharing) {
  if (node < 0) return 0;
  char path[96];
  std::snprintf(path, sizeof path, "/sys/devices/system/node/node%d/cpulist", node);
  FILE *f = std::fopen(path, "r");
  if (!f) return 0;
  char line[4096] = {0};
  const bool ok = std::fgets(line, sizeof line, f) != nullptr;
  std::fclose(f);
  if (!ok) return 0;
  cpu_set_t allowed, set;
  if (sched_getaffinity(0, sizeof allowed, &allowed) != 0) return 0;
  CPU_ZERO(&set);
  unsigned cpus = 0;
  for (char *q = line; *q;) {  // "0-63,128-191"
    char *end = nullptr;
    const long lo = std::strtol(q, &end, 10);
    if (end == q) break;
    long hi = lo;
    if (*end == '-') hi = std::strtol(end + 1, &end, 10);
    for (long c = lo; c <= hi && c < CPU_SETSIZE; ++c)
      if (CPU_ISSET((int)c, &allowed)) CPU_SET((int)c, &set), ++cpus;
    if (*end != ',') break;
    q = end + 1;
  }
  if (!cpus || cpus < threads_sharing) return 0;
  return sched_setaffinity(0, sizeof set, &set) == 0 ? 1 : 0;
}
// NUMA node of the page that holds p (-1: unknown -- not mapped yet, or no such system call)
int hg_numa_node_of(const void *p) {
  void *page = reinterpret_cast<void *>(reinterpret_cast<uintptr_t>(p) & ~(uintptr_t)4095);
  int status = -1;
  const long r = syscall(SYS_move_pages, 0, 1ul, &page, nullptr, &status, 0);  // (nodes == NULL: query only)
  return r == 0 && status >= 0 ? status : -1;
}

// ---- FASTA -----------------------------------------------------------------------------------------------
namespace {
// Merge of FASTA text held in src[0..n) (whole lines; the last one may lack its '\n') into dst: header lines become
// one 'N', sequence lines lose their line ends (src/fastx_reader.rs:14-26).  dst may be src itself: the write index
// never passes the read index.
size_t merge_lines(const uint8_t *src, size_t n, uint8_t *dst) {
  size_t w = 0, i = 0;
  while (i < n) {
    const uint8_t *nl = static_cast<const uint8_t *>(std::memchr(src + i, '\n', n - i));
    const size_t j = nl ? (size_t)(nl - src) : n;
    if (src[i] == '>') {
      dst[w++] = 'N';
    } else {
      size_t e = j;
      if (e > i && src[e - 1] == '\r') --e;  // :19-21 pops '\r' with or without '\n'
      std::memmove(dst + w, src + i, e - i);
      w += e - i;
    }
    i = j < n ? j + 1 : j;
  }
  return w;
}
}  // namespace

namespace {
// needletail's view of the same file (the reference's CPU path, src/sketch.rs:76-87): `parse_fastx_file` picks the
// FASTA or the FASTQ parser from the first byte; a FASTA record's sequence is every line up to the next line that
// starts with '>', a FASTQ record is four lines (@id, sequence, +, qualities); `normalize(false)` then DROPS
// blanks, tabs and line ends inside the sequence (u/U -> T and upper-casing happen on the device, HG_NORM_U2T).
// Output: the read_merge_seq layout ('N' per record start), so that k-mers never span records.
inline bool nt_blank(uint8_t c) { return c == ' ' || c == '\t' || c == '\r' || c == '\n'; }

// first position in [i, n) whose byte is below 0x21 (line ends, blanks, tabs and the other control characters) or,
// read as a signed char, negative; n if there is none.  A sequence line that is clean up to its '\n' is found with
// this ONE scan, 16 bytes per step (all the needletail mode costs over read_merge_seq's memchr).
inline size_t find_ctl(const uint8_t *buf, size_t i, size_t n) {
#if defined(__SSE2__)
  const __m128i lim = _mm_set1_epi8(0x21);
  for (; i + 16 <= n; i += 16) {
    const int m = _mm_movemask_epi8(_mm_cmplt_epi8(_mm_loadu_si128(reinterpret_cast<const __m128i *>(buf + i)), lim));
    if (m) return i + (size_t)__builtin_ctz((unsigned)m);
  }
#endif
  for (; i < n; ++i)
    if (buf[i] < 0x21 || buf[i] >= 0x80) return i;
  return n;
}

// state carried from one block of lines to the next (a file is merged in L2-sized blocks, see hg_read_fastx_impl)
struct NeedletailState {
  bool started = false, fastq = false;
  unsigned line_in_rec = 0;  // FASTQ: 0 = @id, 1 = sequence, 2 = '+', 3 = qualities
};
size_t merge_lines_needletail(const uint8_t *src, size_t n, uint8_t *dst, NeedletailState &st) {
  size_t w = 0, i = 0;
  if (!st.started && n) st.started = true, st.fastq = src[0] == '@';
  const bool fastq = st.fastq;
  unsigned line_in_rec = st.line_in_rec;
  while (i < n) {
    bool is_seq;
    if (fastq) {
      if (line_in_rec == 0) dst[w++] = 'N';
      is_seq = line_in_rec == 1;
      line_in_rec = (line_in_rec + 1) & 3;
    } else {
      is_seq = src[i] != '>';
      if (!is_seq) dst[w++] = 'N';
    }
    size_t j;
    if (is_seq) {
      const size_t t = find_ctl(src, i, n);
      if (t == n || src[t] == '\n') {  // fast path: nothing to drop inside the line
        j = t;
        std::memmove(dst + w, src + i, j - i);
        w += j - i;
      } else {
        const uint8_t *nl = static_cast<const uint8_t *>(std::memchr(src + t, '\n', n - t));
        j = nl ? (size_t)(nl - src) : n;
        for (size_t u = i; u < j; ++u)
          if (!nt_blank(src[u])) dst[w++] = src[u];
      }
    } else {
      const uint8_t *nl = static_cast<const uint8_t *>(std::memchr(src + i, '\n', n - i));
      j = nl ? (size_t)(nl - src) : n;
    }
    i = j < n ? j + 1 : j;
  }
  st.line_in_rec = line_in_rec;
  return w;
}

// default growth: realloc (the buffer is the caller's malloc memory)
bool grow_malloc(uint8_t *&buf, size_t &cap, size_t need, size_t /*keep*/, void *) {
  if (need <= cap) return true;
  uint8_t *nb = static_cast<uint8_t *>(std::realloc(buf, need));
  if (!nb) return false;
  buf = nb, cap = need;
  return true;
}
}  // namespace

// The reader proper; `grow(buf, cap, need, keep, user)` enlarges the buffer keeping its first `keep` bytes
// (realloc here, page-locked memory in hg_api.hip's hg_read_fastx_pinned).
hg_status hg_read_fastx_impl(const char *path, uint32_t mode, uint8_t **pbuf, size_t *pcap, size_t *n_bps,
                             hg_grow_fn grow, void *user) {
  const uint32_t base_mode = mode & 0xFu;
  if (!path || !pbuf || !pcap || !n_bps || base_mode > HG_READ_NEEDLETAIL || (mode & ~(0xFu | HG_READ_PACK2 | HG_READ_PACK2_U2T)))
    return HG_ERR_INVALID;
  *n_bps = 0;
  uint8_t *buf = *pbuf;
  size_t cap = buf ? *pcap : 0;
  const bool pack = (mode & HG_READ_PACK2) != 0, u2t = (mode & HG_READ_PACK2_U2T) != 0;
  const int fd = ::open(path, O_RDONLY | O_CLOEXEC);
  if (fd < 0) return HG_ERR_IO;
  struct stat sb;
  if (::fstat(fd, &sb) != 0 || !S_ISREG(sb.st_mode)) {
    ::close(fd);
    return HG_ERR_IO;
  }
  const size_t fsize = (size_t)sb.st_size;
  // Plain text is merged (and packed) block by block: a block of whole lines is read into a buffer that stays in the
  // core's L2, merged from there into the result -- or into a second small buffer and 2-bit packed into the result.
  // Reading the whole file into the result and merging it in place moved every byte through DRAM four or five times;
  // with 16 reader threads that, not the cores, set the rate (0.6 ms per 5 Mbp file alone, 1.2 ms with 16 running).
  constexpr size_t BLOCK = 256u << 10;
  static thread_local std::vector<uint8_t> blk, mst;
  NeedletailState nst;
  hg_status st = HG_OK;
  size_t w = 0;       // bases (or, unpacked, bytes) produced so far
  size_t carry = 0;   // bytes of an unfinished line at the front of blk
  size_t have = 0;    // packed mode: merged bases waiting in mst for a full group of 32
  bool gzip = false, first = true;
  uint8_t *mask = nullptr;
  try {
    if (blk.size() < BLOCK + 4096) blk.resize(BLOCK + 4096);
    if (pack && mst.size() < blk.size() + 64) mst.resize(blk.size() + 64);
  } catch (const std::bad_alloc &) {
    st = HG_ERR_OOM;
  }
  if (st == HG_OK && !grow(buf, cap, fsize + 64, 0, user)) st = HG_ERR_OOM;
  if (st == HG_OK && pack && !(mask = thread_mask(al16((fsize + 7) / 8) + 64))) st = HG_ERR_OOM;
  // The buffers were sized from fstat(), but the file may hold more than st_size by the time it is read (a file
  // that is still being appended to, procfs-style files that report size 0): every block checks the room it needs
  // first and grows the result -- and, when packing, the mask -- keeping what was produced so far.
  size_t mask_cap = al16((fsize + 7) / 8) + 64;
  auto emit = [&](size_t n_lines_bytes) {  // blk[0, n) holds whole lines (or the file's last, open one)
    // a merged block is never longer than its text: every 'N' replaces a header line of at least one byte
    if (!pack) {
      if (w + n_lines_bytes + 64 > cap && !grow(buf, cap, 2 * (w + n_lines_bytes) + 64, w, user)) {
        st = HG_ERR_OOM;
        return;
      }
      w += base_mode == HG_READ_NEEDLETAIL ? merge_lines_needletail(blk.data(), n_lines_bytes, buf + w, nst)
                                           : merge_lines(blk.data(), n_lines_bytes, buf + w);
      return;
    }
    const size_t bases_max = w + have + n_lines_bytes;  // bases after this block, at most
    if (hg_pack2_size(bases_max) + 64 > cap && !grow(buf, cap, 2 * hg_pack2_size(bases_max) + 64, std::min(cap, (w >> 2) + 16), user)) {
      st = HG_ERR_OOM;
      return;
    }
    if (al16((bases_max + 7) / 8) + 64 > mask_cap) {
      mask_cap = 2 * al16((bases_max + 7) / 8) + 64;
      if (!(mask = thread_mask(mask_cap))) {  // (a resize: the bits collected so far stay)
        st = HG_ERR_OOM;
        return;
      }
    }
    const size_t m = base_mode == HG_READ_NEEDLETAIL ? merge_lines_needletail(blk.data(), n_lines_bytes, mst.data() + have, nst)
                                                     : merge_lines(blk.data(), n_lines_bytes, mst.data() + have);
    const size_t total = have + m, full = total & ~(size_t)31;
    pack2_span(mst.data(), full, u2t, buf + (w >> 2), mask + (w >> 3));  // w is a multiple of 32 here
    w += full;
    have = total - full;
    if (have) std::memmove(mst.data(), mst.data() + full, have);
  };
  while (st == HG_OK) {
    if (blk.size() - carry < BLOCK) {  // a line longer than a block: the block grows with it
      try {
        blk.resize(2 * blk.size());
        if (pack) mst.resize(blk.size() + 64);
      } catch (const std::bad_alloc &) {
        st = HG_ERR_OOM;
        break;
      }
    }
    const ssize_t got = ::read(fd, blk.data() + carry, BLOCK);
    if (got < 0) {
      if (errno == EINTR) continue;
      st = HG_ERR_IO;
      break;
    }
    if (first && got >= 2 && blk[0] == 0x1f && blk[1] == 0x8b) {
      gzip = true;
      break;
    }
    first = false;
    const size_t n = carry + (size_t)got;
    if (got == 0) {  // end of file: what is left is the last line, without its '\n'
      if (n) emit(n);
      break;
    }
    size_t cut = n;  // one past the last '\n'
    while (cut > 0 && blk[cut - 1] != '\n') --cut;
    if (cut) emit(cut);
    carry = n - cut;
    if (carry && cut) std::memmove(blk.data(), blk.data() + cut, carry);
  }
  ::close(fd);
  if (gzip) {
    // gzip: inflate transparently -- what needletail's reader does for the reference's CPU path
    // (src/sketch.rs:76); the reference's GPU reader is plain text only.  Inflated whole, merged in place.
    gzFile f = gzopen(path, "rb");
    if (!f) return HG_ERR_IO;
    gzbuffer(f, 1 << 20);
    size_t n = 0;
    if (!grow(buf, cap, std::max(cap, ((size_t)16 << 20) + 64), 0, user)) st = HG_ERR_OOM;
    int got = 0;
    while (st == HG_OK && (got = gzread(f, buf + n, (unsigned)std::min<size_t>(cap - 64 - n, 1u << 30))) > 0) {
      n += (size_t)got;
      if (n == cap - 64 && !grow(buf, cap, 2 * cap, n, user)) st = HG_ERR_OOM;
    }
    if (st == HG_OK && got < 0) st = HG_ERR_IO;
    gzclose(f);
    *pbuf = buf, *pcap = cap;
    if (st != HG_OK) return st;
    NeedletailState gst;
    w = base_mode == HG_READ_NEEDLETAIL ? merge_lines_needletail(buf, n, buf, gst) : merge_lines(buf, n, buf);
    std::memset(buf + w, 0, 64);
    *n_bps = w;
    if (pack) return hg_pack2(buf, w, u2t ? HG_NORM_U2T : HG_NORM_ACGT, buf);  // hg_pack2_size(w) <= w + 64 <= cap
    return HG_OK;
  }
  *pbuf = buf, *pcap = cap;  // the (possibly moved) buffer stays the caller's, also on error
  if (st != HG_OK) return st;
  if (pack) {
    pack2_span(mst.data(), have, u2t, buf + (w >> 2), mask + (w >> 3));
    w += have;
    pack2_finish(buf, w, mask);  // hg_pack2_size(w) + 64 <= cap: emit() made the room
  } else {
    std::memset(buf + w, 0, 64);
  }
  *n_bps = w;
  return HG_OK;
}

extern "C" hg_status hg_read_fastx_into(const char *path, uint32_t mode, uint8_t **pbuf, size_t *pcap, size_t *n_bps) {
  return hg_read_fastx_impl(path, mode, pbuf, pcap, n_bps, grow_malloc, nullptr);
}

extern "C" hg_status hg_read_merge_seq_into(const char *path, uint8_t **pbuf, size_t *pcap, size_t *n_bps) {
  return hg_read_fastx_into(path, HG_READ_MERGE, pbuf, pcap, n_bps);
}

extern "C" hg_status hg_read_merge_seq(const char *path, uint8_t **out, size_t *n_bps) {
  if (!path || !out || !n_bps) return HG_ERR_INVALID;
  *out = nullptr, *n_bps = 0;
  uint8_t *buf = nullptr;
  size_t cap = 0;
  const hg_status st = hg_read_merge_seq_into(path, &buf, &cap, n_bps);
  if (st != HG_OK) {
    std::free(buf);
    return st;
  }
  *out = buf;
  return HG_OK;
}

extern "C" void hg_free(void *p) { std::free(p); }
