// ThreadSanitizer driver (CPU build) for the host threads of a host-fed call: CallPool + hg_pack2_piece
// (hyper-gen_amd/csrc/hg_host.h, hg_formats.cpp) -- the way hg_sketch_batch packs a sub-batch -- against hg_pack2.
#include <cstdio>
#include <cstring>
#include <random>
#include <utility>
#include <vector>

#include "../../hyper-gen_amd/csrc/hg_host.h"

int main() {
  std::mt19937_64 rng(7);
  const char alpha[] = "ACGTacgtNnUuRY-";
  const size_t lens[] = {0, 1, 63, 64, 65, 1000, (1u << 20) - 1, 1u << 20, (1u << 20) + 1, 3 * (1u << 20) + 77, 2500000};
  std::vector<std::vector<uint8_t>> seqs;
  for (size_t n : lens) {
    std::vector<uint8_t> s(n);
    for (auto &c : s) c = (uint8_t)alpha[rng() % (sizeof alpha - 1)];
    if (n > 300) std::memset(s.data() + n / 3, 'N', 200);
    seqs.push_back(std::move(s));
  }
  for (unsigned threads : {1u, 2u, 7u}) {
    CallPool pool(threads);
    for (int round = 0; round < 3; ++round) {  // the pool is reused sub-batch after sub-batch
      for (uint32_t norm : {HG_NORM_ACGT, HG_NORM_U2T}) {
        std::vector<size_t> off;
        size_t total = 0;
        for (auto &s : seqs) off.push_back(total), total += hg_pack2_size(s.size());
        std::vector<uint8_t> got(total + 1, 0xEE), want(total + 1, 0xEE);
        constexpr size_t PIECE = 1u << 20;
        std::vector<std::pair<size_t, size_t>> pieces;
        for (size_t g = 0; g < seqs.size(); ++g)
          for (size_t b = 0; b < seqs[g].size(); b += PIECE) pieces.emplace_back(g, b);
        pool.run(pieces.size(), [&](size_t i) {
          const size_t g = pieces[i].first, b = pieces[i].second, n = seqs[g].size();
          hg_pack2_piece(seqs[g].data(), n, norm, got.data() + off[g], b, b + PIECE < n ? b + PIECE : n);
        });
        pool.run(0, [&](size_t) { std::abort(); });
        for (size_t g = 0; g < seqs.size(); ++g)
          if (hg_pack2(seqs[g].data(), seqs[g].size(), norm, want.data() + off[g]) != HG_OK) return 2;
        if (got != want) {
          std::printf("pieces differ from hg_pack2 (threads %u, norm %u)\n", threads, norm);
          return 3;
        }
      }
    }
  }
  std::printf("tsan driver ok\n");
  return 0;
}
