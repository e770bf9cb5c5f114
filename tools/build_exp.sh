#!/bin/bash
# Development build of the library with the k-mer kernel's A/B switch compiled in (-DHG_KMER_EXPERIMENT: the
# K = 21 canonical kernel variant is then chosen by the HG_KMER_VARIANT environment variable; the product build
# reads no environment).  Output: tools/_exp_libhypergen.so (git-ignored); use it with HYPERGEN_LIB=...
set -e
ROOT=$(cd "$(dirname "$0")/.." && pwd)
C=$ROOT/hyper-gen_amd/csrc
make -s -j6 -C "$C" all
hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -Wall -Wno-unused-function -ffp-contract=off \
  -mllvm -amdgpu-atomic-optimizer-strategy=None -DHG_KMER_EXPERIMENT "$@" -c "$C/hg_kmer_kernels.hip" -o /tmp/_exp_kmer.o
hipcc --offload-arch=gfx950 -shared -fPIC -o "$ROOT/tools/_exp_libhypergen.so" /tmp/_exp_kmer.o \
  $(ls "$C"/*.o | grep -v hg_kmer_kernels.o) -lz -lpthread
echo "built tools/_exp_libhypergen.so"
