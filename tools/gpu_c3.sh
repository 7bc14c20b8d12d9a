#!/bin/bash
# C3 (progressive, trellis, optimized scans) A/B: parity tests of the progressive path, then bench --config c3 with the
# given environment settings, one line each.  usage: gpu_c3.sh TAG "ENV1=.. ENV2=.." "ENV.." ...
cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/${1:-c3}; shift; mkdir -p "$O"
timeout 600 python -m pytest tests/test_gpu_parity.py tests/test_gpu_fuzz.py -x -q -k "prog or scan or c3 or fuzz or golden" 2>&1 | tail -3
i=0
for envs in "$@"; do
  i=$((i+1))
  env $envs timeout 300 python bench.py --config c3 --no-cpu-baseline --no-host-leg --no-inflight-leg > "$O/bench_$i.log" 2>&1
  python - "$O/bench_$i.log" "$envs" <<'PY'
import json,sys
try:
    d=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1]); r=d['roofline']
    print(sys.argv[2], '|', d['ms_per_step'], d['value'], d['bit_exact'])
    print('   ', r['kernel_ms_per_call(untimed pass, every kernel bracketed)'])
except Exception as e:
    print(sys.argv[1], 'ERR', e); print(open(sys.argv[1]).read()[-600:])
PY
done
