#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/r4b; mkdir -p "$O"
timeout 300 tools/probes/valu_rate > "$O/valu_rate.jsonl" 2>&1; echo "valu_rate rc=$?"; grep -c op "$O/valu_rate.jsonl"
timeout 1500 python -m pytest tests -x -q -m gpu -n 4 -p no:cacheprovider 2>&1 | tail -4
MJH_GUARD=3 timeout 1500 python -m pytest tests -q -m gpu -n 4 -p no:cacheprovider > "$O/suite_mode3.log" 2>&1; echo "mode 3 rc=$?"; tail -3 "$O/suite_mode3.log"
for c in c5 c5t metric; do
  timeout 400 python bench.py --config $c --no-cpu-baseline --no-host-leg --no-inflight-leg > "$O/bench_$c.log" 2>&1
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
