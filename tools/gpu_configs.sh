#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/${1:-configs}; mkdir -p "$O"
for c in c3 c2 c4 c5 c5t; do
  timeout 400 python bench.py --config $c --no-cpu-baseline --no-host-leg --no-inflight-leg > "$O/bench_$c.log" 2>&1
  python - "$O/bench_$c.log" <<'PY'
import json,sys
try:
    d=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1]); r=d['roofline']
    print(d['config']['config_key'], d['ms_per_step'], d['value'], d['bit_exact'], r['kernel'], r['kernel_ms'], r['frac'])
    print('   ', r['kernel_ms_per_call(untimed pass, every kernel bracketed)'])
except Exception as e:
    print(sys.argv[1], 'ERR', e); print(open(sys.argv[1]).read()[-600:])
PY
done
