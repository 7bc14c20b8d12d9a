#!/bin/bash
# arithmetic configurations at 256 frames per step: first under the memory fence (offsets beyond 2^31 elements are new
# territory: 256 x 12.4 M coefficients), then the plain bench lines
cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/arith256; mkdir -p "$O"
MJH_GUARD=2 timeout 600 python bench.py --config arith --batch 256 --warmup 1 --steps 1 --no-host-leg --no-inflight-leg --no-cpu-baseline > "$O/guard_arith.log" 2>&1
echo "guarded rc=$?"; tail -c 400 "$O/guard_arith.log" | grep -o '"bit_exact": {[^}]*}' 
for c in arith arith_prog; do
  timeout 600 python bench.py --config $c --batch 256 --warmup 1 --steps 2 --no-host-leg --no-inflight-leg --cpu-budget 10 > "$O/bench_$c.log" 2>&1
  python - "$O/bench_$c.log" <<'PY'
import json,sys
try:
    d=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1]); r=d['roofline']
    print(d['config'].get('config_key'), d['ms_per_step'], 'ms/step', d['value'], 'Mpx/s', d['bit_exact'], 'cpu', d.get('cpu_baseline',{}).get('value'))
    print('   ', r['kernel_ms_per_call(untimed pass, every kernel bracketed)'])
except Exception as e:
    print(sys.argv[1], 'ERR', e); print(open(sys.argv[1]).read()[-800:])
PY
done
