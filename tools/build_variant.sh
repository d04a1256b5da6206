#!/bin/bash
# tools/build_variant.sh NAME [extra hipcc flags...] -> tools/dbg/NAME.so (experiments; load with RUSTRACER_AMD_SO)
set -e
cd "$(dirname "$0")/../rustracer_amd/csrc"
NAME=$1; shift
mkdir -p ../../tools/dbg
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=off -fno-fast-math \
  -fhip-fp32-correctly-rounded-divide-sqrt -fno-gpu-flush-denormals-to-zero -fno-slp-vectorize -Wall -Wno-unused-function "$@" \
  -shared -o ../../tools/dbg/$NAME.so rt_kernels.hip
