// Host-only seams between hg_formats.cpp (no HIP) and hg_api.hip.
#pragma once
#include <cstddef>
#include <cstdint>

#include "../../include/hypergen.h"

// enlarge `buf` to at least `need` bytes keeping its first `keep` bytes; false = out of memory
typedef bool (*hg_grow_fn)(uint8_t *&buf, size_t &cap, size_t need, size_t keep, void *user);

hg_status hg_read_fastx_impl(const char *path, uint32_t mode, uint8_t **pbuf, size_t *pcap, size_t *n_bps,
                             hg_grow_fn grow, void *user);
