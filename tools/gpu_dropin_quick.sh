#!/bin/bash
# drop-in throughput of an unchanged libjpeg client (tests/native/mt_bench) through the preloaded shim, with the shim's own phase timing
cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/${1:-dropin}; mkdir -p "$O"
export LD_LIBRARY_PATH=$PWD/oracle/_ref:$LD_LIBRARY_PATH
for t in 1 4 16; do
  n=$((t == 1 ? 40 : 16))
  MOZJPEG_HIP_TIMING=1 LD_PRELOAD=$PWD/mozjpeg_amd/libmozjpeg_hip_jpeg62.so timeout 300 tests/native/mt_bench $t $n 3840 2160 75 baseline > "$O/mt_$t.json" 2> "$O/mt_$t.err"
  python -c "
import json,sys
d=json.loads(open('$O/mt_$t.json').read().strip().splitlines()[-1]); print('threads',d['threads'],'images/s',d['images_per_s'],'Mpx/s',d['mpix_per_s'],'hash',d['fnv1a_first'])"
  grep timing "$O/mt_$t.err"
done
timeout 600 python -m pytest tests/test_gpu_dropin.py tests/test_gpu_host_path.py -x -q > "$O/pytest.log" 2>&1; tail -3 "$O/pytest.log"
