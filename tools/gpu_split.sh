#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/${1:-split}; mkdir -p "$O"
for dc in 0 1; do
  MJH_SPLIT_DC=$dc timeout 300 python tools/bench_variants.py --env MJH_SPLIT --variants 1,2,3,4,6 --steps 20 > "$O/split_dc$dc.log" 2>&1
  echo "MJH_SPLIT_DC=$dc"; grep "^{" "$O/split_dc$dc.log" | cut -c1-120
done
GPU_MAX_HW_QUEUES=8 MJH_SPLIT_DC=0 timeout 300 python tools/bench_variants.py --env MJH_SPLIT --variants 3,4 --steps 20 > "$O/split_hwq8.log" 2>&1
echo "GPU_MAX_HW_QUEUES=8 dc0"; grep "^{" "$O/split_hwq8.log" | cut -c1-120
timeout 600 python -m pytest tests/test_gpu_host_path.py tests/test_gpu_parity.py -x -q -k "split or ordered or full_size" > "$O/pytest.log" 2>&1; tail -3 "$O/pytest.log"
