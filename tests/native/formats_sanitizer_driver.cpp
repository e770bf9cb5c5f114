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
  printf("asan driver ok\n");
  return 0;
}
