#!/bin/bash
# one GPU call: targeted tests, the default bench line, and the 2-rank launch path on one GPU (gloo switch)
cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/${1:-round}; mkdir -p "$O"
timeout 900 python -m pytest tests/test_gpu_host_path.py tests/test_gpu_parity.py -x -q -k "split or ordered or full_size or pipelined" > "$O/pytest.log" 2>&1; tail -3 "$O/pytest.log"
timeout 600 python bench.py > "$O/bench_default.log" 2>&1; tail -1 "$O/bench_default.log" | cut -c1-1500
MJH_BENCH_DIST_BACKEND=gloo timeout 600 python bench.py --gpus 2 --steps 5 --warmup 3 --batch 16 --host-seconds 1 --verify 2 > "$O/bench_2ranks_gloo.log" 2>&1; tail -1 "$O/bench_2ranks_gloo.log" | cut -c1-1800
