#!/bin/bash
# Round 5: two small A/Bs on the metric workload: which DC chains run late (MJH_DC_LATE 0 / 1 / 2), grid of the general trellis tier
cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/r5j; mkdir -p "$O"
timeout 300 python tools/bench_variants.py --env MJH_DC_LATE --variants 1,2,0,1,2 --steps 20 > "$O/dclate.log" 2>&1; grep '^{' "$O/dclate.log" | cut -c1-400
for g in 2048 4096 8192 1024; do MJH_QD_GRID=$g timeout 200 python tools/bench_variants.py --env MJH_NOP --variants 0 --steps 20 > "$O/qd_$g.log" 2>&1; echo "-- qd grid $g"; grep '^{' "$O/qd_$g.log" | cut -c1-400; done
