#!/bin/bash
# drop-in throughput A/B over environment settings of the shim; usage: gpu_dropin_ab.sh TAG "ENV=.." "ENV=.." ...
cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/${1:-dropin_ab}; shift; mkdir -p "$O"
export LD_LIBRARY_PATH=$PWD/oracle/_ref:$LD_LIBRARY_PATH
for envs in "$@"; do
  echo "== $envs"
  for t in 1 4 16; do
    n=$((t == 1 ? 40 : 16))
    env $envs MOZJPEG_HIP_TIMING=1 LD_PRELOAD=$PWD/mozjpeg_amd/libmozjpeg_hip_jpeg62.so timeout 300 tests/native/mt_bench $t $n 3840 2160 75 baseline > "$O/mt.json" 2> "$O/mt.err"
    python -c "
import json,sys
d=json.loads(open('$O/mt.json').read().strip().splitlines()[-1]); print('threads',d['threads'],'images/s',d['images_per_s'],'Mpx/s',d['mpix_per_s'],'hash',d['fnv1a_first'])"
    grep timing "$O/mt.err"
  done
done
