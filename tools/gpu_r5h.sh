#!/bin/bash
# Round 5: A/B of the FDCT kernel's shape (sets of 64 blocks per wave, prefetch of the next set's rows, forced occupancy):
# the same sources built with different -D flags into gpu_variants/lib_*.so, each timed on the metric workload
cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/r5h; mkdir -p "$O"
for v in nb1 nb4 nb4pf nb4pf3 nb2pf3 nb8 nb1; do
  MOZJPEG_AMD_LIB=$PWD/gpu_variants/lib_$v.so timeout 200 python tools/bench_variants.py --env MJH_NOP --variants 0 --steps 10 > "$O/$v.log" 2>&1
  echo "-- $v"; grep '^{' "$O/$v.log" | cut -c1-420; grep -i "error\|fault\|Traceback" "$O/$v.log" | head -3
done
for v in nb4 nb4pf3; do
  echo "-- parity $v"; MOZJPEG_AMD_LIB=$PWD/gpu_variants/lib_$v.so timeout 300 python -m pytest tests/test_gpu_parity.py -q -m gpu -x -k "full_size or every_stage" 2>&1 | tail -2
done
