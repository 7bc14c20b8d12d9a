#!/bin/bash
# a library variant for same-box A/B runs (MOZJPEG_AMD_LIB): mjh_kernels.hip recompiled with extra flags, the other objects as built
# usage: bash tools/build_variant.sh NAME [-DFLAG ...]   ->  mozjpeg_amd/variants/libmozjpeg_hip_NAME.so
set -e
cd "$(dirname "$0")/.."
NAME=$1; shift
mkdir -p mozjpeg_amd/variants
C=mozjpeg_amd/csrc
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off -fPIC -fno-fast-math -Wall -Wno-unused-function "$@" -x hip -c $C/mjh_kernels.hip -o mozjpeg_amd/variants/mjh_kernels_$NAME.o
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o mozjpeg_amd/variants/libmozjpeg_hip_$NAME.so mozjpeg_amd/variants/mjh_kernels_$NAME.o $C/mjh_prog.o $C/mjh_arith.o $C/mjh_encoder.o $C/mjh_pool.o $C/mjh_guard.o $C/mjh_numa.o
rm -f mozjpeg_amd/variants/mjh_kernels_$NAME.o
echo mozjpeg_amd/variants/libmozjpeg_hip_$NAME.so
