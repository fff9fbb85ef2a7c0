#!/bin/bash
# A library with the kernels of ONE regime only (gendr_capi.hip GENDR_DEV_MIN: opt_shape.py's renderers + the team kernels): compiles in
# seconds, for A/B runs of those kernels (tools/ab_shape.sh).   bash tools/devbuild.sh v_name.so [-DGENDR_...=N ...]
ROOT=$(cd "$(dirname "$0")/.." && pwd)
# GENDR_DEV_MIN=2 in the environment: BASELINE config 2's kernels instead (12 s), e.g. for compiler-flag A/B runs with tools/ab.sh:
#   GENDR_DEV_MIN=2 bash tools/devbuild.sh x_maxilp.so -mllvm -amdgpu-sched-strategy=max-ilp
OUT=$1; shift
hipcc --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off -fPIC -shared -munsafe-fp-atomics -fno-slp-vectorize -DGENDR_DEV_MIN=${GENDR_DEV_MIN:-1} "$@" \
      "$ROOT/gendr_amd/csrc/gendr_capi.hip" -o "$ROOT/$OUT"
