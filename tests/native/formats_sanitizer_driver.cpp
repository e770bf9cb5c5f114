#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <random>
#include <string>
#include <vector>
#include <zlib.h>
#include "hypergen.h"
// usage: driver <scratch-dir>   (built with -fsanitize=address,undefined by tests/test_host_sanitizers.py)
int main(int argc, char **argv) {
  const std::string T = argc > 1 ? argv[1] : "/tmp";
  std::mt19937_64 rng(5);
  // pack / unpack round trips for every q and several sizes
  for (int rep = 0; rep < 200; ++rep) {
    uint32_t d = 256 * (1 + rng() % 8);
    int amp = 1 << (rng() % 15);
    std::vector<int16_t> hv(d), back(d);
    for (auto &x : hv) x = (int16_t)((int64_t)(rng() % (2 * amp)) - amp);
    uint32_t q = hg_hv_quant_bits(hv.data(), d);
    std::vector<uint8_t> packed((size_t)q * d / 8);
    if (hg_hv_pack(hv.data(), d, q, packed.data()) != HG_OK) return 1;
    if (hg_hv_unpack(packed.data(), d, q, back.data()) != HG_OK) return 2;
    if (q < 16 && memcmp(hv.data(), back.data(), d * 2)) { printf("roundtrip mismatch q=%u\n", q); return 3; }
  }
  // the naive (non-AVX2) payload layout: exact-size buffers, every q, hv_d not a multiple of 8 / 16 / 256; the round trip
  // is lossless except where the reference's is not (-2^(q-1) comes back positive; q = 16): those values are avoided
  for (int rep = 0; rep < 300; ++rep) {
    const uint32_t d = 1 + (uint32_t)(rng() % 1200), q = 2 + (uint32_t)(rng() % 14);  // q in 2..15
    std::vector<int16_t> hv(d), back(d);
    const int lim = (1 << (q - 1)) - 1;  // values in [-lim, lim]
    for (auto &x : hv) x = (int16_t)((int64_t)(rng() % (2 * (uint64_t)lim + 1)) - lim);
    const size_t nb = hg_hv_packed_bytes_naive(d, q);
    if (nb != 2 * (((size_t)q * d + 16) / 16)) return 30;
    std::vector<uint8_t> packed(nb);  // exact size: ASan sees any byte written or read behind it
    if (hg_hv_pack_naive(hv.data(), d, q, packed.data()) != HG_OK) return 31;
    if (hg_hv_unpack_naive(packed.data(), d, q, back.data()) != HG_OK) return 32;
    if (memcmp(hv.data(), back.data(), (size_t)d * 2)) { printf("naive roundtrip mismatch q=%u d=%u\n", q, d); return 33; }
    if (hg_hv_payload_layout(d, q, nb) != HG_PAYLOAD_NAIVE) return 34;
    if (d % 256 == 0 && hg_hv_payload_layout(d, q, hg_hv_packed_bytes(d, q)) != HG_PAYLOAD_BITPACKER8X) return 35;
    if (hg_hv_payload_layout(d, q, nb + 2) != -1) return 36;
  }
  for (uint32_t q : {1u, 16u}) {  // the lossy corners only have to stay inside their buffers
    const uint32_t d = 333;
    std::vector<int16_t> hv(d), back(d);
    for (auto &x : hv) x = (int16_t)rng();
    std::vector<uint8_t> packed(hg_hv_packed_bytes_naive(d, q));
    if (hg_hv_pack_naive(hv.data(), d, q, packed.data()) != HG_OK || hg_hv_unpack_naive(packed.data(), d, q, back.data()) != HG_OK) return 37;
  }
  if (hg_hv_pack_naive(nullptr, 8, 9, nullptr) == HG_OK || hg_hv_unpack_naive(nullptr, 8, 0, nullptr) == HG_OK) return 38;
  // sketch file write / read
  std::vector<hg_file_sketch> recs(5);
  std::vector<std::vector<int16_t>> pay(5);
  std::vector<std::string> names(5);
  for (int i = 0; i < 5; ++i) {
    pay[i].resize(100 * (i + 1));
    for (auto &x : pay[i]) x = (int16_t)rng();
    names[i] = "/some/path/genome_" + std::to_string(i) + ".fna";
    memset(&recs[i], 0, sizeof recs[i]);
    recs[i].ksize = 21, recs[i].canonical = 1, recs[i].hv_quant_bits = 9, recs[i].hv_norm_2 = 1234 + i;
    recs[i].scaled = 1500, recs[i].seed = 123, recs[i].hv_d = 4096, recs[i].file_str = names[i].c_str();
    recs[i].hv = pay[i].data(), recs[i].hv_len = pay[i].size();
  }
  if (hg_sketch_file_write((T + "/t.sketch").c_str(), recs.data(), 5) != HG_OK) return 4;
  hg_sketch_file *f = nullptr;
  if (hg_sketch_file_read((T + "/t.sketch").c_str(), &f) != HG_OK) return 5;
  for (size_t i = 0; i < hg_sketch_file_count(f); ++i) {
    const hg_file_sketch *r = hg_sketch_file_get(f, i);
    if (r->hv_len != pay[i].size() || memcmp(r->hv, pay[i].data(), r->hv_len * 2) || names[i] != r->file_str) return 6;
  }
  hg_sketch_file_free(f);
  // the image reader (no payload copies): same records, payload bytes found at their offsets inside the image
  {
    hg_sketch_file *g = nullptr;
    if (hg_sketch_file_read_image((T + "/t.sketch").c_str(), &g) != HG_OK) return 40;
    size_t img_bytes = 0;
    const uint8_t *img = hg_sketch_file_image(g, &img_bytes);
    if (!img || hg_sketch_file_count(g) != 5) return 41;
    for (size_t i = 0; i < 5; ++i) {
      const hg_file_sketch *r = hg_sketch_file_get(g, i);
      const uint64_t off = hg_sketch_file_payload_offset(g, i);
      if (r->hv != nullptr || r->hv_len != pay[i].size() || off + r->hv_len * 2 > img_bytes) return 42;
      if (memcmp(img + off, pay[i].data(), r->hv_len * 2) || names[i] != r->file_str) return 43;
    }
    hg_sketch_file_free(g);
    // the copying reader has no image
    hg_sketch_file *h = nullptr;
    if (hg_sketch_file_read((T + "/t.sketch").c_str(), &h) != HG_OK) return 44;
    size_t nb = 1;
    if (hg_sketch_file_image(h, &nb) != nullptr || nb != 0) return 45;
    hg_sketch_file_free(h);
  }
  // truncated / corrupt files must fail cleanly
  for (long cut : {0L, 7L, 8L, 20L, 60L, 200L}) {
    FILE *in = fopen((T + "/t.sketch").c_str(), "rb"); std::vector<char> all(1 << 16); size_t n = fread(all.data(), 1, all.size(), in); fclose(in);
    FILE *out = fopen((T + "/c.sketch").c_str(), "wb"); fwrite(all.data(), 1, (size_t)std::min<long>(cut, (long)n), out); fclose(out);
    hg_sketch_file *g = nullptr;
    hg_status st = hg_sketch_file_read((T + "/c.sketch").c_str(), &g);
    if (st == HG_OK) hg_sketch_file_free(g);
  }
  // random corruption of length fields / payload: must fail or succeed, never read out of bounds
  {
    FILE *in = fopen((T + "/t.sketch").c_str(), "rb"); std::vector<unsigned char> all(1 << 16); size_t n = fread(all.data(), 1, all.size(), in); fclose(in);
    for (int rep = 0; rep < 300; ++rep) {
      std::vector<unsigned char> bad(all.begin(), all.begin() + n);
      for (int k = 0; k < 1 + (int)(rng() % 4); ++k) bad[rng() % n] = (unsigned char)rng();
      FILE *out = fopen((T + "/c.sketch").c_str(), "wb"); fwrite(bad.data(), 1, n, out); fclose(out);
      hg_sketch_file *g = nullptr;
      if (hg_sketch_file_read((T + "/c.sketch").c_str(), &g) == HG_OK) {
        for (size_t i = 0; i < hg_sketch_file_count(g); ++i) {
          const hg_file_sketch *r = hg_sketch_file_get(g, i);
          volatile unsigned acc = 0;
          for (uint64_t j = 0; j < r->hv_len; ++j) acc += (unsigned)r->hv[j];
          acc += (unsigned)strlen(r->file_str);
        }
        hg_sketch_file_free(g);
      }
      g = nullptr;
      if (hg_sketch_file_read_image((T + "/c.sketch").c_str(), &g) == HG_OK) {
        size_t img_bytes = 0;
        const uint8_t *img = hg_sketch_file_image(g, &img_bytes);
        for (size_t i = 0; i < hg_sketch_file_count(g); ++i) {
          const hg_file_sketch *r = hg_sketch_file_get(g, i);
          const uint64_t off = hg_sketch_file_payload_offset(g, i);
          if (off > img_bytes || r->hv_len * 2 > img_bytes - off) return 46;  // an accepted record lies inside the image
          volatile unsigned acc = 0;
          for (uint64_t j = 0; j < r->hv_len * 2; ++j) acc += img[off + j];
          acc += (unsigned)strlen(r->file_str);
        }
        hg_sketch_file_free(g);
      }
    }
  }
  // FASTA readers: plain, CRLF, no trailing newline, empty, gz, reuse buffer
  const char *cases[] = {">a\nACGT\nAC\n>b\nGG\n", ">a\r\nACGT\r\nAC\r\n", ">x\nACGT", "", "\n\n", ">only header\n"};
  uint8_t *buf = nullptr; size_t cap = 0;
  for (const char *c : cases) {
    FILE *o = fopen((T + "/t.fa").c_str(), "wb"); fwrite(c, 1, strlen(c), o); fclose(o);
    uint8_t *p = nullptr; size_t n = 0;
    if (hg_read_merge_seq((T + "/t.fa").c_str(), &p, &n) != HG_OK) return 7;
    size_t n2 = 0;
    if (hg_read_merge_seq_into((T + "/t.fa").c_str(), &buf, &cap, &n2) != HG_OK) return 8;
    if (n != n2 || (n && memcmp(p, buf, n))) return 9;
    gzFile z = gzopen((T + "/t.fa.gz").c_str(), "wb"); gzwrite(z, c, (unsigned)strlen(c)); gzclose(z);
    uint8_t *pz = nullptr; size_t nz = 0;
    if (hg_read_merge_seq((T + "/t.fa.gz").c_str(), &pz, &nz) != HG_OK) return 10;
    if (nz != n || (n && memcmp(p, pz, n))) return 11;
    hg_free(p); hg_free(pz);
  }
  hg_free(buf);
  // block-wise reader (256 KiB blocks of whole lines), both read modes, 2-bit packing (also in place), long single line
  {
    std::string big = ">rec one\n";
    unsigned x = 12345;
    const char alpha[] = "ACGTacgtNnUu \t";
    for (int line = 0; line < 9000; ++line) {
      const int len = 1 + (int)(x % 97);
      for (int i = 0; i < len; ++i) { x = x * 1664525u + 1013904223u; big += alpha[(x >> 24) % (line % 50 ? 8 : 14)]; }
      big += (line % 7) ? "\n" : "\r\n";
      if (line % 1500 == 1499) big += ">next record\n";
    }
    big += ">long\n";
    for (int i = 0; i < 700000; ++i) { x = x * 1664525u + 1013904223u; big += alpha[(x >> 24) % 4]; }  // no trailing newline
    FILE *o = fopen((T + "/big.fa").c_str(), "wb"); fwrite(big.data(), 1, big.size(), o); fclose(o);
    for (uint32_t mode : {HG_READ_MERGE, HG_READ_NEEDLETAIL}) {
      uint8_t *a = nullptr, *b = nullptr; size_t ca = 0, cb = 0, na = 0, nb = 0;
      if (hg_read_fastx_into((T + "/big.fa").c_str(), mode, &a, &ca, &na) != HG_OK) return 12;
      const uint32_t norm = mode == HG_READ_NEEDLETAIL ? HG_NORM_U2T : HG_NORM_ACGT;
      const uint32_t flags = mode | HG_READ_PACK2 | (norm == HG_NORM_U2T ? HG_READ_PACK2_U2T : 0u);
      if (hg_read_fastx_into((T + "/big.fa").c_str(), flags, &b, &cb, &nb) != HG_OK) return 13;
      if (na != nb || na < 700000) return 14;
      std::vector<uint8_t> blob(hg_pack2_size(na));
      if (hg_pack2(a, na, norm, blob.data()) != HG_OK) return 15;
      if (memcmp(blob.data(), b, blob.size())) return 16;
      std::vector<uint8_t> inplace(std::max(na, hg_pack2_size(na)));
      memcpy(inplace.data(), a, na);
      if (hg_pack2(inplace.data(), na, norm, inplace.data()) != HG_OK) return 17;
      if (memcmp(blob.data(), inplace.data(), blob.size())) return 18;
      hg_free(a); hg_free(b);
    }
    for (size_t n : {(size_t)0, (size_t)1, (size_t)31, (size_t)32, (size_t)33, (size_t)4097}) {
      std::vector<uint8_t> q(n ? n : 1, 'A'), out(hg_pack2_size(n) + 1, 0x5A);
      if (hg_pack2(q.data(), n, HG_NORM_ACGT, out.data()) != HG_OK || out[hg_pack2_size(n)] != 0x5A) return 19;
    }
  }
  // the sparse 2-bit form (codes + run table): exact-capacity buffers, runs that cross and end on 64-base words, in place,
  // and the capacity answer; the table must describe the same positions as hg_pack2's bitmap
  for (int rep = 0; rep < 200; ++rep) {
    const size_t n = rng() % 3000;
    std::vector<uint8_t> seq(n ? n : 1);
    const char base[] = "ACGTacgt";
    for (size_t i = 0; i < n; ++i) seq[i] = (uint8_t)base[rng() % 8];
    for (int k = 0, runs = (int)(rng() % 6); k < runs && n; ++k) {
      size_t st = rng() % n, len = 1 + rng() % (rep % 3 ? 200 : 5);
      if (rep % 7 == 0) st = st / 64 * 64, len = 64 * (1 + rng() % 3);  // whole words
      for (size_t i = st; i < std::min(n, st + len); ++i) seq[i] = 'N';
    }
    if (rep % 11 == 0 && n) seq[n - 1] = 'N';
    const size_t cb = (((n + 3) / 4) + 15) / 16 * 16;
    std::vector<uint8_t> dense(hg_pack2_size(n) ? hg_pack2_size(n) : 1);
    if (hg_pack2(seq.data(), n, HG_NORM_ACGT, dense.data()) != HG_OK) return 50;
    size_t need = 0;
    std::vector<uint8_t> tiny(cb + 16);  // room for the header only: more than zero runs must answer HG_ERR_CAPACITY
    hg_status st = hg_pack2s(seq.data(), n, HG_NORM_ACGT, tiny.data(), tiny.size(), &need);
    if (st != HG_OK && st != HG_ERR_CAPACITY) return 51;
    if (st == HG_ERR_CAPACITY && need <= tiny.size()) return 52;
    std::vector<uint8_t> blob(need);  // exactly what it asked for
    size_t got = 0;
    if (hg_pack2s(seq.data(), n, HG_NORM_ACGT, blob.data(), blob.size(), &got) != HG_OK || got != need) return 53;
    if (memcmp(blob.data(), dense.data(), cb)) return 54;
    uint32_t n_runs; memcpy(&n_runs, blob.data() + cb, 4);
    if (hg_pack2s_size(n, n_runs) != need) return 55;
    std::vector<uint8_t> mask((n + 7) / 8 + 1, 0);
    uint64_t prev_end = 0;
    for (uint32_t r = 0; r < n_runs; ++r) {
      uint32_t a, len; memcpy(&a, blob.data() + cb + 8 + 8 * r, 4); memcpy(&len, blob.data() + cb + 12 + 8 * r, 4);
      if (len == 0 || (r && a <= prev_end) || (uint64_t)a + len > n) return 56;  // maximal, ascending, inside the sequence
      prev_end = (uint64_t)a + len;
      for (uint32_t i = a; i < a + len; ++i) mask[i >> 3] |= (uint8_t)(1u << (i & 7));
    }
    if (memcmp(mask.data(), dense.data() + cb, (n + 7) / 8)) return 57;
    std::vector<uint8_t> inplace(std::max(n, need) ? std::max(n, need) : 1);
    if (n) memcpy(inplace.data(), seq.data(), n);
    if (hg_pack2s(inplace.data(), n, HG_NORM_ACGT, inplace.data(), inplace.size(), &got) != HG_OK || got != need) return 58;
    if (memcmp(inplace.data(), blob.data(), need)) return 59;
  }
  // a file that holds more than fstat() said (procfs reports st_size 0; the same happens to a FASTA that is appended
  // to while it is read): the result must equal that of a regular copy of the same bytes, in every mode
  {
    std::vector<char> all;
    FILE *in = fopen("/proc/cpuinfo", "rb");
    if (in) {
      char tmp[4096]; size_t g;
      while ((g = fread(tmp, 1, sizeof tmp, in)) > 0) all.insert(all.end(), tmp, tmp + g);
      fclose(in);
    }
    if (all.size() > 64) {
      FILE *o = fopen((T + "/cpuinfo.copy").c_str(), "wb"); fwrite(all.data(), 1, all.size(), o); fclose(o);
      for (uint32_t mode : {(uint32_t)HG_READ_MERGE, (uint32_t)HG_READ_NEEDLETAIL, (uint32_t)(HG_READ_MERGE | HG_READ_PACK2),
                            (uint32_t)(HG_READ_NEEDLETAIL | HG_READ_PACK2 | HG_READ_PACK2_U2T)}) {
        uint8_t *a = nullptr, *b = nullptr; size_t ca = 0, cb = 0, na = 0, nb = 0;
        if (hg_read_fastx_into("/proc/cpuinfo", mode, &a, &ca, &na) != HG_OK) return 20;
        if (hg_read_fastx_into((T + "/cpuinfo.copy").c_str(), mode, &b, &cb, &nb) != HG_OK) return 21;
        // (the clock readings inside /proc/cpuinfo may differ between two reads: lengths of the merged text need not)
        if (na == 0 || nb == 0) return 22;
        const size_t bytes = (mode & HG_READ_PACK2) ? hg_pack2_size(na) : na;
        volatile unsigned acc = 0;
        for (size_t i = 0; i < bytes; ++i) acc += a[i];
        hg_free(a); hg_free(b);
      }
      // a caller-provided buffer that is far too small (the recycled-slot case)
      uint8_t *small = (uint8_t *)malloc(64); size_t cs = 64, ns = 0;
      if (hg_read_fastx_into("/proc/cpuinfo", HG_READ_MERGE, &small, &cs, &ns) != HG_OK || ns == 0 || cs < ns + 64) return 23;
      hg_free(small);
    }
  }
  printf("asan driver ok\n");
  return 0;
}
