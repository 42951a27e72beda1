#!/bin/bash
# Build a library that CONTAINS the retired experiment kernels and the one image-changing knob (never the product build):
#   trace_kernel_bvh2 (two paths per lane, round 3), trace_kernel_bvhx (walker / shader waves, round 4), RAYZ_DEBUG_CHUNK_CAP.
#   bash tools/build_experiments.sh [shader waves of the exchange kernel: 4 (default), 5, 6] [extra hipcc flags, e.g. -DRAYZ_BVH_PROFILE]
# -> variants/lib_experiments_s<N>.so ; run a tool against it with:  bash tools/with_lib.sh variants/lib_experiments_s4.so python tools/bvhx_bench.py
set -e
cd "$(dirname "$0")/.."
S=${1:-4}; shift || true
mkdir -p variants
hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=off -fno-gpu-flush-denormals-to-zero -fhip-fp32-correctly-rounded-divide-sqrt \
      -DRAYZ_EXPERIMENTS -DRAYZ_BVHX_SHADERS=$S "$@" -shared -o variants/lib_experiments_s$S.so rayz_amd/csrc/rayz_hip.hip rayz_amd/host/rayz_host.cpp
echo variants/lib_experiments_s$S.so
