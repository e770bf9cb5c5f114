// hg_sketch_plan.hip -- the geometry of a sketch batch: per-genome hit regions and the work items of the k-mer kernel, the
// plan the ctx keeps between calls, its upload, parameter checks, device-side 2-bit packing of an ASCII batch.
// What src/sketch.rs:35-56 gets for free from one task per file: here n genomes share one launch.
#include <algorithm>
#include <cstring>

#include "hg_sketch.h"

static uint32_t round_cap(uint64_t cap) {
  if (cap > HG_SORT_LDS_MAX_KEYS) {  // in-place global sort needs a power of two
    uint64_t p = 1;
    while (p < cap) p <<= 1;
    cap = p;
  }
  return cap > 0xFFFFFFF0ull ? 0xFFFFFFF0u : (uint32_t)cap;
}

hg_status hg_check_sketch_params(hg_ctx *c, const hg_sketch_params *p) {
  if (!p) return hg_fail(c, HG_ERR_INVALID, "params == NULL");
  if (p->ksize < 1) return hg_fail(c, HG_ERR_INVALID, "ksize must be >= 1");
  if (p->ksize > 255) return hg_fail(c, HG_ERR_UNSUPPORTED, "ksize must be <= 255 (the reference's -k is u8)");
  if (p->scaled < 1) return hg_fail(c, HG_ERR_INVALID, "scaled must be >= 1");
  if (p->hv_layout > HG_LAYOUT_AVX2 || p->norm_mode > HG_NORM_U2T) return hg_fail(c, HG_ERR_INVALID, "bad layout / norm mode");
  if (p->hv_d == 0 || p->hv_d > 32768) return hg_fail(c, HG_ERR_UNSUPPORTED, "hv_d must be in 1..32768");
  return HG_OK;
}

hg_status hg_plan_build(hg_ctx *c, const uint64_t *offsets, const uint64_t *lens, size_t n, uint32_t ksize, uint64_t scaled,
                        const std::vector<uint32_t> *want_caps, hg_batch_tables &t, const uint64_t *mask_offs) {
  const uint64_t item_starts = hg_kmer_item_starts(ksize);
  // groups of work items (one workgroup each): consecutive items of SMALL genomes are put together while the group stays
  // within hg_kmer_item_tiles tiles; an item that fills a work item on its own (a piece of a large genome) stays alone, so
  // that the launch of large genomes keeps its granularity
  const uint32_t tile_starts = hg_kmer_tile_starts(ksize), item_tiles = hg_kmer_item_tiles(ksize);
  const uint32_t full_item_tiles = tile_starts ? (uint32_t)((item_starts + tile_starts - 1) / tile_starts) : 0;
  bool group_open = false;  // the last group may take more items
  t.group_first.clear();
  uint32_t group_tiles = 0;
  t.meta.resize(n);
  t.item_genome.clear();
  uint64_t slot = 0, max_expect = 0;
  uint32_t max_cap = 0;
  for (size_t g = 0; g < n; ++g) {
    if (offsets[g] & 3) return hg_fail(c, HG_ERR_INVALID, "genome offsets must be multiples of 4");
    hg_genome_meta &m = t.meta[g];
    m.seq_off = offsets[g];
    m.n_bps = lens[g];
    m.mask_off = mask_offs ? mask_offs[g] : offsets[g] + (((lens[g] + 3) / 4 + 15) & ~(uint64_t)15);  // (read by the packed kernels only)
    const uint64_t n_starts = lens[g] >= ksize ? lens[g] - ksize + 1 : 0;
    const uint64_t expect = n_starts / scaled;
    uint64_t cap = expect * 2 + 1024;    // expected n_starts/scaled; sd ~ sqrt of that
    if (cap > n_starts) cap = n_starts;  // can never exceed the number of k-mers
    if (want_caps && (*want_caps)[g] > cap) cap = (*want_caps)[g];
    if (cap == 0) cap = 1;
    m.hit_cap = round_cap(cap);
    m.hit_off = slot;
    slot += m.hit_cap;
    max_cap = std::max(max_cap, m.hit_cap);
    max_expect = std::max(max_expect, std::min(expect, n_starts));
    const uint64_t n_items = (n_starts + item_starts - 1) / item_starts;
    if (t.item_genome.size() + n_items > 0x7FFFFFFFull)
      return hg_fail(c, HG_ERR_UNSUPPORTED, "batch too large for one launch; split it");
    m.item_first = (uint32_t)t.item_genome.size();
    if (tile_starts) {
      for (uint64_t it = 0; it < n_items; ++it) {
        const uint64_t starts = std::min<uint64_t>(item_starts, n_starts - it * item_starts);
        const uint32_t tiles = (uint32_t)((starts + tile_starts - 1) / tile_starts);
        const bool alone = tiles >= full_item_tiles;
        if (!group_open || alone || group_tiles + tiles > item_tiles) {
          t.group_first.push_back((uint32_t)(t.item_genome.size() + it));
          group_tiles = 0;
        }
        group_tiles += tiles;
        group_open = !alone;
      }
    }
    t.item_genome.insert(t.item_genome.end(), (size_t)n_items, (uint32_t)g);
  }
  t.n_groups = t.group_first.size();
  if (t.n_groups) t.group_first.push_back((uint32_t)t.item_genome.size());
  t.total_slots = slot;
  t.max_cap = max_cap;
  t.max_expect = (uint32_t)std::min<uint64_t>(max_expect, 0xFFFFFFF0ull);
  t.n_items = t.item_genome.size();
  return HG_OK;
}

bool hg_plan_matches(const hg_ctx *c, const uint64_t *offsets, const uint64_t *lens, const uint64_t *mask_offs, size_t n,
                     uint32_t ksize, uint64_t scaled, bool packed) {
  const hg_sketch_plan *pl = c->plan.get();
  if (!pl || pl->ksize != ksize || pl->scaled != scaled || pl->packed != packed || pl->offs.size() != n) return false;
  if (std::memcmp(pl->offs.data(), offsets, n * 8) != 0 || std::memcmp(pl->lens.data(), lens, n * 8) != 0) return false;
  if (mask_offs) return pl->masks.size() == n && std::memcmp(pl->masks.data(), mask_offs, n * 8) == 0;
  return pl->masks.empty();
}

void hg_plan_tables_from_cache(const hg_sketch_plan &pl, size_t n, hg_batch_tables &t, bool with_meta) {
  t.total_slots = pl.total_slots, t.max_cap = pl.max_cap, t.max_expect = pl.max_expect, t.n_items = pl.n_items;
  t.n_groups = pl.n_groups;
  t.item_genome.clear();
  t.group_first.clear();
  if (!with_meta) {
    t.meta.clear();
    return;
  }
  t.meta.resize(n);
  uint64_t slot = 0;
  for (size_t g = 0; g < n; ++g) {  // only what callers read back: capacities and hit offsets
    t.meta[g].hit_cap = pl.caps[g];
    t.meta[g].hit_off = slot;
    slot += pl.caps[g];
  }
}

hg_status hg_plan_upload(hg_ctx *c, const hg_batch_tables &t, const uint64_t *offsets, const uint64_t *lens,
                         const uint64_t *mask_offs, size_t n, uint32_t ksize, uint64_t scaled, bool packed) {
  c->plan.reset();  // (the device tables are about to change)
  hg_status s;
  const size_t n_items = t.item_genome.size();
  if ((s = hg_ensure(c, c->w_gmeta, n * sizeof(hg_genome_meta) + 16)) != HG_OK) return s;
  const size_t items_words = t.n_groups ? hg_plan_group_offset(n_items) + t.n_groups + 1 : n_items;  // items, then the group table
  if ((s = hg_ensure(c, c->w_items, items_words * sizeof(uint32_t) + 16)) != HG_OK) return s;
  const size_t pin_meta = (n * sizeof(hg_genome_meta) + 63) & ~(size_t)63;
  const size_t need = pin_meta + items_words * sizeof(uint32_t) + 64;
  // the staging area may still feed the previous plan's upload (a sync-free step returns with it queued): that upload
  // lies in front of that step's kernels, so this waits for the step BEFORE the previous one at most
  if (c->plan_upload_pending) {
    HG_HIP(c, hipEventSynchronize(c->plan_uploaded));
    c->plan_upload_pending = false;
  }
  if (need > c->h_plan_cap) {
    if (c->h_plan) (void)hipHostFree(c->h_plan);
    c->h_plan = nullptr, c->h_plan_cap = 0;
    const size_t want = need + need / 4 + 4096;
    hipError_t e = hipHostMalloc(&c->h_plan, want, hipHostMallocDefault);
    if (e != hipSuccess) return hg_fail(c, HG_ERR_OOM, std::string("hipHostMalloc: ") + hipGetErrorString(e));
    c->h_plan_cap = want;
  }
  if (!c->plan_uploaded) HG_HIP(c, hipEventCreateWithFlags(&c->plan_uploaded, hipEventDisableTiming));
  auto *pin = static_cast<uint8_t *>(c->h_plan);
  std::memcpy(pin, t.meta.data(), n * sizeof(hg_genome_meta));
  HG_HIP(c, hipMemcpyAsync(c->w_gmeta.p, pin, n * sizeof(hg_genome_meta), hipMemcpyHostToDevice, c->stream));
  if (n_items) {
    std::memcpy(pin + pin_meta, t.item_genome.data(), n_items * sizeof(uint32_t));
    if (t.n_groups) {
      std::memset(pin + pin_meta + n_items * sizeof(uint32_t), 0, (hg_plan_group_offset(n_items) - n_items) * sizeof(uint32_t));
      std::memcpy(pin + pin_meta + hg_plan_group_offset(n_items) * sizeof(uint32_t), t.group_first.data(), (t.n_groups + 1) * sizeof(uint32_t));
    }
    HG_HIP(c, hipMemcpyAsync(c->w_items.p, pin + pin_meta, items_words * sizeof(uint32_t), hipMemcpyHostToDevice, c->stream));
  }
  HG_HIP(c, hipEventRecord(c->plan_uploaded, c->stream));
  c->plan_upload_pending = true;
  auto pl = std::make_shared<hg_sketch_plan>();
  pl->offs.assign(offsets, offsets + n);
  pl->lens.assign(lens, lens + n);
  if (mask_offs) pl->masks.assign(mask_offs, mask_offs + n);
  pl->caps.resize(n);
  for (size_t g = 0; g < n; ++g) pl->caps[g] = t.meta[g].hit_cap;
  pl->ksize = ksize, pl->scaled = scaled, pl->packed = packed;
  pl->total_slots = t.total_slots, pl->max_cap = t.max_cap, pl->max_expect = t.max_expect, pl->n_items = n_items;
  pl->n_groups = t.n_groups;
  c->plan = std::move(pl);
  return HG_OK;
}

hg_status hg_pack_batch(hg_ctx *c, const uint8_t *d_seq, const uint64_t *seq_offs, const uint64_t *lens, size_t n,
                        uint32_t norm_mode, uint8_t *d_blobs, const uint64_t *blob_offs) {
  if (n == 0) return HG_OK;
  hg_status s;
  if ((s = hg_ensure(c, c->w_pktab, 3 * n * sizeof(uint64_t) + 64)) != HG_OK) return s;
  if ((s = hg_ensure_pinned(c, 3 * n * sizeof(uint64_t) + 64)) != HG_OK) return s;
  HG_HIP(c, hipStreamSynchronize(c->stream));  // (the scratch may still feed an earlier upload)
  auto *tab = static_cast<uint64_t *>(c->h_pin);
  uint64_t max_len = 0;
  for (size_t g = 0; g < n; ++g) {
    if ((seq_offs[g] & 3) || (blob_offs[g] & 15)) return hg_fail(c, HG_ERR_INVALID, "pack2: sequence offsets must be multiples of 4, blob offsets of 16");
    tab[3 * g] = seq_offs[g], tab[3 * g + 1] = lens[g], tab[3 * g + 2] = blob_offs[g];
    max_len = std::max(max_len, lens[g]);
  }
  HG_HIP(c, hipMemcpyAsync(c->w_pktab.p, tab, 3 * n * sizeof(uint64_t), hipMemcpyHostToDevice, c->stream));
  const uint64_t groups = ((((max_len + 7) / 8 + 15) & ~(uint64_t)15) + 3) / 4;  // lanes per genome: one per 4 bitmap bytes
  const uint64_t blocks = (groups + 255) / 256;
  if (blocks > 0x7FFFFFFFull) return hg_fail(c, HG_ERR_UNSUPPORTED, "pack2: genome too long for one launch");
  if (blocks)
    HG_HIP(c, hg_launch_pack2(c->stream, d_seq, static_cast<const uint64_t *>(c->w_pktab.p), (uint32_t)n, (uint32_t)blocks,
                              norm_mode == HG_NORM_U2T ? 1u : 0u, d_blobs));
  HG_HIP(c, hipStreamSynchronize(c->stream));  // (the pinned table is free again)
  return HG_OK;
}

// The plan of a batch as numbers (host only, no device involved): for tests and for callers that want to size a batch before
// they submit it.  counts[0..5] = work items, workgroups of the k-mer launch, hit slots, largest hit region, largest expected
// sampled count, tiles per full work item; group_first (optional, cap entries) receives the first work item of every workgroup
// followed by the item count.
extern "C" hg_status hg_sketch_plan_describe(const uint64_t *offsets, const uint64_t *lens, size_t n, uint32_t ksize, uint64_t scaled,
                                             uint64_t counts[6], uint32_t *group_first, size_t cap) {
  if ((n && (!offsets || !lens)) || !counts || ksize < 1 || ksize > 255 || scaled < 1) return HG_ERR_INVALID;
  hg_batch_tables t;
  const hg_status s = hg_plan_build(nullptr, offsets, lens, n, ksize, scaled, nullptr, t, nullptr);
  if (s != HG_OK) return s;
  const uint32_t tile = hg_kmer_tile_starts(ksize);
  counts[0] = t.n_items, counts[1] = t.n_groups ? t.n_groups : t.n_items, counts[2] = t.total_slots, counts[3] = t.max_cap;
  counts[4] = t.max_expect, counts[5] = tile ? (hg_kmer_item_starts(ksize) + tile - 1) / tile : 0;
  if (group_first) {
    if (t.group_first.size() > cap) return HG_ERR_CAPACITY;
    for (size_t i = 0; i < t.group_first.size(); ++i) group_first[i] = t.group_first[i];
  }
  return HG_OK;
}
