#!/bin/bash
# one GPU call: A/B of the tile-sorted AC trellis (MJH_TRELLIS_V3 = passes per tile, 0 = general kernel) + parity tests
cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/${1:-v3}; mkdir -p "$O"
timeout 400 python tools/bench_variants.py --env MJH_TRELLIS_V3 --variants ${2:-0,4,1,2,8} --steps 10 > "$O/variants.log" 2>&1
tail -6 "$O/variants.log" | cut -c1-600
timeout ${3:-900} python -m pytest tests/test_gpu_parity.py tests/test_gpu_large.py -x -q > "$O/parity.log" 2>&1
tail -5 "$O/parity.log"
