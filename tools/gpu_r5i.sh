#!/bin/bash
# Round 5: A/B of MJH_DC_LATE (the chroma DC chains + final DC statistics behind the AC kernel instead of next to it) on the
# metric / C2 / C5t workloads, then the whole suite on the tree (FDCT kernel: four sets per wave)
cd "$GRAFT_REPO_ROOT" || exit 1
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r5i; mkdir -p "$O"
ab() { timeout 300 python tools/bench_variants.py --config $1 --env MJH_DC_LATE --variants 0,1,0,1 --steps $2 > "$O/ab_$1.log" 2>&1; echo "-- $1"; grep '^{' "$O/ab_$1.log" | cut -c1-460; grep -i "error\|fault\|Traceback" "$O/ab_$1.log" | head -3; }
echo "== 1. A/B"; ab metric 10; ab c2 10; ab c5t 5
echo "== 2. the whole suite"; timeout 700 python -m pytest tests -q -m gpu -x > "$O/suite.log" 2>&1; tail -3 "$O/suite.log"
echo "== 3. bench"; timeout 400 python bench.py > "$O/bench.log" 2>&1; tail -1 "$O/bench.log" | cut -c1-330
