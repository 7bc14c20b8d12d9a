#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/arith; mkdir -p "$O"
[ -n "$ARITH_SKIP_TESTS" ] || timeout 900 python -m pytest tests -q -m gpu -k "arith" -n 4 -p no:cacheprovider 2>&1 | tail -3
for c in ${ARITH_CONFIGS:-arith arith_prog}; do
  timeout 900 python bench.py --config $c --warmup 1 --steps ${ARITH_STEPS:-2} --no-host-leg --no-inflight-leg --cpu-budget 15 > "$O/bench_$c.log" 2>&1
  python - "$O/bench_$c.log" <<'PY'
import json,sys
try:
    d=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1]); r=d['roofline']
    print(d['config'].get('config_key'), d['ms_per_step'], 'ms/step', d['value'], 'Mpx/s', d['bit_exact'], 'cpu', d.get('cpu_baseline',{}).get('value'), d.get('cpu_baseline',{}).get('cores'))
    print('   ', r['kernel_ms_per_call(untimed pass, every kernel bracketed)'])
except Exception as e:
    print(sys.argv[1], 'ERR', e); print(open(sys.argv[1]).read()[-800:])
PY
done
