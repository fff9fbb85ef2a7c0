#!/bin/bash
# Builds the library from the working tree with extra flags into a file at the repo root (for tools/ab.sh):
#   bash tools/variant.sh v_name.so [-DGENDR_...=N ...]
ROOT=$(cd "$(dirname "$0")/.." && pwd)
OUT=$1; shift
hipcc --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off -fPIC -shared -munsafe-fp-atomics -fno-slp-vectorize "$@" \
      "$ROOT/gendr_amd/csrc/gendr_capi.hip" -o "$ROOT/$OUT"
