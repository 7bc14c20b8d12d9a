#!/bin/bash
# Round 5: timing of the metric / C3 / C5t workloads + the whole GPU suite + SQ counter passes, for a kernel edit
# usage: gpurun --timeout 1000 -- 'bash tools/gpu_r5e.sh TAG'
TAG=${1:-r5e}
cd "$GRAFT_REPO_ROOT" || exit 1
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=gpurun_out/$TAG; mkdir -p "$O"
t() { timeout 300 python tools/bench_variants.py --config $1 --env MJH_NOP --variants 0,0 --steps $2 > "$O/t_$1.log" 2>&1; echo "-- $1"; grep '^{' "$O/t_$1.log" | cut -c1-560; grep -i "error\|fault\|Traceback" "$O/t_$1.log" | head -3; }
echo "== 1. kernel times"; t metric 10; t c3 5; t c5t 5
echo "== 2. the whole suite"; timeout 600 python -m pytest tests -q -m gpu -x > "$O/suite.log" 2>&1; tail -3 "$O/suite.log"
echo "== 3. SQ counters"; bash tools/gpu_sq.sh $TAG/sq 2>&1 | tail -8 | cut -c1-250
