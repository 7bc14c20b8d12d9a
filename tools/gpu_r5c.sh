#!/bin/bash
# Round 5, third GPU call: the round-synchronous trellis walk (k_trellis_ac_v3 and the general tiers) after the removal of the
# round-4 variants that lost their A/B -- timing on the metric / C3 / C5t workloads (r5a has the numbers before), the whole GPU
# suite, the bench line; then what PC-sampling configurations this rocprofv3 offers on the device.
cd "$GRAFT_REPO_ROOT" || exit 1
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r5c; mkdir -p "$O"
t() { timeout 300 python tools/bench_variants.py --config $1 --env MJH_NOP --variants 0,0 --steps $2 > "$O/t_$1.log" 2>&1; echo "-- $1"; grep '^{' "$O/t_$1.log" | cut -c1-560; grep -i "error\|fault\|Traceback" "$O/t_$1.log" | head -3; }
echo "== 1. kernel times"; t metric 10; t c3 5; t c5t 5; t c2 10
echo "== 2. the whole suite"; timeout 600 python -m pytest tests -q -m gpu -x > "$O/suite.log" 2>&1; tail -3 "$O/suite.log"
echo "== 3. bench"; timeout 400 python bench.py > "$O/bench.log" 2>&1; tail -1 "$O/bench.log" | cut -c1-330
echo "== 4. PC sampling configurations"; timeout 60 rocprofv3 -L > "$O/avail.txt" 2>&1; grep -n -i -B2 -A12 "pc.sampl" "$O/avail.txt" | head -60
