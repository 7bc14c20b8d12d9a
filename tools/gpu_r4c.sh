#!/bin/bash
cd /tmp && export TMPDIR=/tmp
cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/r4c; mkdir -p "$O"
timeout 600 tools/probes/valu_rate > "$O/valu_rate.jsonl" 2>&1; echo "valu_rate rc=$?"; grep -c op "$O/valu_rate.jsonl"
timeout 1500 python -m pytest tests -x -q -m gpu -n 4 -p no:cacheprovider 2>&1 | tail -4
Q="--no-cpu-baseline --no-host-leg --no-inflight-leg"
for c in c5t c3 metric c2; do
  timeout 400 python bench.py --config $c $Q > "$O/bench_$c.log" 2>&1
  python - "$O/bench_$c.log" <<'PY'
import json,sys
try:
    d=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1]); r=d['roofline']
    print(d['config'].get('config_key'), d['ms_per_step'], d['value'], d['bit_exact']['ok'], r['kernel'], r['kernel_ms'], r['frac'])
    print('   ', r['kernel_ms_per_call(untimed pass, every kernel bracketed)'])
except Exception as e:
    print(sys.argv[1], 'ERR', e); print(open(sys.argv[1]).read()[-600:])
PY
done
timeout 200 rocprofv3 --kernel-trace --stats -d "$O" -o c5t_stats -- python bench.py --config c5t --steps 5 --warmup 2 $Q --verify 1 > "$O/c5t_stats.log" 2>&1
python tools/rocprof_summary.py "$(find $O -name 'c5t_stats_results.db' | head -1)" > "$O/c5t_kernel_stats.csv"; head -8 "$O/c5t_kernel_stats.csv"
