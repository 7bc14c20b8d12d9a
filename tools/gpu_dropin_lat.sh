#!/bin/bash
# one client thread, unchanged libjpeg client (tests/native/mt_bench), 4K: images/s and the shim's phase timing; A/B over MJH_DC_SPEC
cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/${1:-dropin_lat}; mkdir -p "$O"
export LD_LIBRARY_PATH=$PWD/oracle/_ref:$LD_LIBRARY_PATH
for spec in 1 0; do
  for t in 1 4; do
    MJH_DC_SPEC=$spec MOZJPEG_HIP_TIMING=1 LD_PRELOAD=$PWD/mozjpeg_amd/libmozjpeg_hip_jpeg62.so timeout 300 tests/native/mt_bench $t $((t == 1 ? 1500 : 600)) 3840 2160 75 baseline > "$O/mt.json" 2> "$O/mt.err"
    python -c "
import json,sys
d=json.loads(open('$O/mt.json').read().strip().splitlines()[-1]); print('MJH_DC_SPEC=$spec threads',d['threads'],'images',d['images'],'seconds',d['seconds'],'images/s',d['images_per_s'],'hash',d['fnv1a_first'])"
    grep timing "$O/mt.err"
  done
done
