#!/bin/bash
# the whole GPU suite, then the contract bench and the C3 bench (short forms); usage: gpu_suite.sh TAG
cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/${1:-suite}; mkdir -p "$O"
timeout 1500 python -m pytest tests -x -q -m gpu 2>&1 | tail -4
timeout 300 python bench.py --no-cpu-baseline --no-host-leg --no-inflight-leg > "$O/bench_c1.log" 2>&1
timeout 300 python bench.py --config c3 --no-cpu-baseline --no-host-leg --no-inflight-leg > "$O/bench_c3.log" 2>&1
for f in "$O/bench_c1.log" "$O/bench_c3.log"; do python - "$f" <<'PY'
import json,sys
try:
    d=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1]); r=d['roofline']
    print(d['config'].get('config_key'), d['ms_per_step'], d['value'], d['bit_exact']['ok'], r['kernel'], r['kernel_ms'], r['frac'])
    print('   ', r['kernel_ms_per_call(untimed pass, every kernel bracketed)'])
except Exception as e:
    print(sys.argv[1], 'ERR', e); print(open(sys.argv[1]).read()[-600:])
PY
done
