#!/bin/bash
# bash tools/build_variant.sh <name> [-D... flags]  ->  ./ab_<name>.so at the repo root (git-ignored; travels with gpurun)
NAME=$1; shift
cd /root/repo/gendr_amd/csrc && /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off -fPIC -shared -munsafe-fp-atomics -fno-slp-vectorize "$@" gendr_capi.hip -o /root/repo/ab_$NAME.so 2>&1 | grep -E "error" ; ls -la /root/repo/ab_$NAME.so | cut -c30-
