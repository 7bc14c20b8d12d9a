#!/bin/bash
# Round 5, second GPU call: K2 with the interleaved histogram copies (timing + parity subset), then PC sampling of the metric
# workload (rocprofv3 beta feature: own process, short timeout, last in the call).
cd "$GRAFT_REPO_ROOT" || exit 1
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r5b; mkdir -p "$O"
echo "== 1. metric workload, kernel times (K2 histogram copies interleaved)"; timeout 300 python tools/bench_variants.py --env MJH_NOP --variants 0,0 --steps 10 > "$O/k2hist.log" 2>&1; grep '^{' "$O/k2hist.log" | cut -c1-520
echo "== 2. parity subset"; timeout 400 python -m pytest tests/test_gpu_parity.py -q -m gpu -x -k "every_stage or full_size or q_opt" > "$O/parity.log" 2>&1; tail -2 "$O/parity.log"
echo "== 3. PC sampling"; bash tools/gpu_pcsample.sh ${1:-stochastic} 2>&1 | tail -80
