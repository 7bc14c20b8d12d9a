#!/bin/bash
# one 4K frame per encode call: per-interval HIP-event times, pure kernel durations (rocprofv3) and the drop-in's client rate
cd /tmp && export TMPDIR=/tmp
cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/${1:-single}; mkdir -p "$O"
Q="--no-cpu-baseline --no-host-leg --no-inflight-leg --verify 1"
timeout 300 python bench.py --batch 1 --steps 300 $Q > "$O/bench_b1.log" 2>&1
python - "$O/bench_b1.log" <<'PY'
import json,sys
d=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1]); r=d['roofline']
print('batch 1:', d['ms_per_step'], 'ms per frame;', d['value'], 'Mpx/s')
t=r['kernel_ms_per_call(untimed pass, every kernel bracketed)']; print('   sum of intervals %.3f' % sum(v for k,v in t.items() if 'side' not in k)); print('   ', t)
PY
timeout 200 rocprofv3 --kernel-trace --stats -d "$O" -o b1 -- python bench.py --batch 1 --steps 50 --warmup 5 $Q > "$O/b1_stats.log" 2>&1
python tools/rocprof_summary.py "$(find $O -name 'b1_results.db' | head -1)" > "$O/b1_kernel_stats.csv"; cat "$O/b1_kernel_stats.csv" | head -40
