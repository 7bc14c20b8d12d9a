#!/bin/bash
# Round 5: traffic passes of C2 and C5t on the final tree (their focus interval is the AC trellis, whose scan loop changed last)
cd "$GRAFT_REPO_ROOT" || exit 1
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r05x; mkdir -p "$O"
Q="--no-cpu-baseline --no-host-leg --no-inflight-leg --verify 1 --other-configs none"
for c in c2 c5t; do
  timeout 60 rocprofv3 --pmc FETCH_SIZE --kernel-trace -d "$O" -o ${c}_fetch -- python bench.py --config $c --steps 2 --warmup 1 $Q > "$O/${c}_fetch.log" 2>&1
  timeout 60 rocprofv3 --pmc WRITE_SIZE --kernel-trace -d "$O" -o ${c}_write -- python bench.py --config $c --steps 2 --warmup 1 $Q > "$O/${c}_write.log" 2>&1
done
ls "$O"
