#!/bin/bash
# two batches in flight inside the encoder: the metric line per setting; usage: gpu_inflight.sh TAG [config]
TAG=${1:-r06if}; CFG=${2:-metric}
cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/$TAG; mkdir -p "$O"
Q="--no-cpu-baseline --no-host-leg --no-inflight-leg --other-configs none --verify 2"
for v in ${VARIANTS:-"MJH_INFLIGHT=1" "MJH_INFLIGHT_MODE=0" "MJH_INFLIGHT_MODE=1" "MJH_INFLIGHT_MODE=2"}; do
  env $v timeout 300 python bench.py --config $CFG $Q > "$O/b.log" 2>&1
  python - "$O/b.log" "$v" <<'PY'
import json,sys
try:
    d=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1]); r=d['roofline']
    print(sys.argv[2], '| ms', d['ms_per_step'], 'value', d['value'], 'ok', d['bit_exact']['ok'], r['kernel'], r['kernel_ms'], 'frac', r['frac'])
    print('    ', {k:v for k,v in list(r['kernel_ms_per_call(untimed pass, every kernel bracketed)'].items())[:8]})
except Exception as e:
    print(sys.argv[2], 'ERR', e); print(open(sys.argv[1]).read()[-800:])
PY
done
