#!/bin/bash
# Round 5: the cleaned-up tree on the chip -- whole GPU suite, the suite's parity file under the memory fence (MJH_GUARD=2:
# an unmapped page behind every device buffer), kernel times of the BASELINE configurations, the bench line.
cd "$GRAFT_REPO_ROOT" || exit 1
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r5f; mkdir -p "$O"
echo "== 1. the whole suite"; timeout 700 python -m pytest tests -q -m gpu -x > "$O/suite.log" 2>&1; tail -3 "$O/suite.log"
echo "== 2. parity + host path under MJH_GUARD=2"; MJH_GUARD=2 timeout 600 python -m pytest tests/test_gpu_parity.py tests/test_gpu_host_path.py -q -m gpu -x > "$O/guard2.log" 2>&1; tail -3 "$O/guard2.log"
t() { timeout 300 python tools/bench_variants.py --config $1 --env MJH_NOP --variants 0 --steps $2 > "$O/t_$1.log" 2>&1; echo "-- $1"; grep '^{' "$O/t_$1.log" | cut -c1-560; grep -i "error\|fault\|Traceback" "$O/t_$1.log" | head -3; }
echo "== 3. kernel times"; t metric 10; t c3 5; t c5 5
echo "== 4. bench"; timeout 400 python bench.py > "$O/bench.log" 2>&1; tail -1 "$O/bench.log" | cut -c1-330
