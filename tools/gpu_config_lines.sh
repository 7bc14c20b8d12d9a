#!/bin/bash
# bench lines of the other configurations once their own PMC traffic passes are committed (roofline.traffic is looked up
# per configuration); usage: gpu_config_lines.sh TAG
cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/${1:-cfg}; mkdir -p "$O"
for c in c3 c2 c4 c5 c5t; do
  timeout 400 python bench.py --config $c --cpu-budget 10 --no-host-leg > "$O/bench_$c.log" 2>&1; tail -1 "$O/bench_$c.log" | cut -c1-200
done
