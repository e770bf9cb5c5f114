#!/bin/bash
# Development build of the library with extra flags for ONE translation unit (A/B switches of the kernels).
# usage: tools/build_variant.sh <name> <file.hip> [flags...]   ->   tools/_exp_lib_<name>.so (git-ignored); run with HYPERGEN_LIB=...
set -e
ROOT=$(cd "$(dirname "$0")/.." && pwd)
C=$ROOT/hyper-gen_amd/csrc
NAME=$1; FILE=$2; shift 2
EXTRA=""
[ "$FILE" = "hg_kmer_kernels.hip" ] && EXTRA="-mllvm -amdgpu-atomic-optimizer-strategy=None"
hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -Wall -Wno-unused-function -ffp-contract=off $EXTRA "$@" -c "$C/$FILE" -o /tmp/_exp_$NAME.o
hipcc --offload-arch=gfx950 -shared -fPIC -o "$ROOT/tools/_exp_lib_$NAME.so" /tmp/_exp_$NAME.o \
  $(ls "$C"/*.o | grep -v "${FILE%.hip}.o") -lz -lpthread
echo "built tools/_exp_lib_$NAME.so"
