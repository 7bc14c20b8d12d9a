#!/bin/bash
# one GPU call: A/B of compiled library variants (MOZJPEG_AMD_LIB) on a configuration, alternating; every process also
# A/Bs an environment knob
# usage: bash tools/gpu_libab.sh TAG CONFIG BATCH ENVNAME VARIANTS lib1 lib2 ...   ("default" = the in-tree library)
cd "$GRAFT_REPO_ROOT" || exit 1
TAG=$1; CFG=$2; BATCH=$3; ENVN=$4; VARS=$5; shift 5
O=gpurun_out/$TAG; mkdir -p "$O"
for rep in 1 2; do
  for lib in "$@"; do
    if [ "$lib" = default ]; then unset MOZJPEG_AMD_LIB; else export MOZJPEG_AMD_LIB="$PWD/$lib"; fi
    echo "== $lib (rep $rep)"
    timeout 400 python tools/bench_variants.py --config $CFG --batch $BATCH --env "$ENVN" --variants "$VARS" --steps 60 --prof 2 2>&1 | tee -a "$O/$(basename $lib).log" | grep '^{' | cut -c1-260
  done
done
