#!/bin/bash
# a library variant for same-box A/B runs (MOZJPEG_AMD_LIB): ONE kernel file recompiled with extra flags, the other objects as built
# usage: bash tools/build_variant.sh NAME UNIT [-DFLAG | -mllvm ... ]   ->  mozjpeg_amd/variants/libmozjpeg_hip_NAME.so   (UNIT: mjh_kernels | mjh_trellis | mjh_prog | mjh_arith)
set -e
cd "$(dirname "$0")/.."
NAME=$1; UNIT=$2; shift 2
mkdir -p mozjpeg_amd/variants
C=mozjpeg_amd/csrc
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off -fPIC -fno-fast-math -Wall -Wno-unused-function "$@" -x hip -c $C/$UNIT.hip -o mozjpeg_amd/variants/${UNIT}_$NAME.o
OBJS=""
for u in mjh_kernels mjh_trellis mjh_prog mjh_arith mjh_encoder mjh_pool mjh_guard mjh_numa; do
  if [ $u = $UNIT ]; then OBJS="$OBJS mozjpeg_amd/variants/${UNIT}_$NAME.o"; else OBJS="$OBJS $C/$u.o"; fi
done
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o mozjpeg_amd/variants/libmozjpeg_hip_$NAME.so $OBJS
rm -f mozjpeg_amd/variants/${UNIT}_$NAME.o
echo mozjpeg_amd/variants/libmozjpeg_hip_$NAME.so
