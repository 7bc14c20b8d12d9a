#!/bin/bash
# Memory-checker runs on the GPU box.  usage: gpu_guard.sh TAG "MODES" [pytest args]   (MODES e.g. "2" or "2 3 1"; "probe" runs the self-test)
cd "$GRAFT_REPO_ROOT" || exit 1
TAG=${1:-g}; MODES=${2:-2}; shift 2
O=gpurun_out/guard_$TAG; mkdir -p "$O"
for m in $MODES; do
  if [ "$m" = probe ]; then
    timeout 900 python tools/guard_probe.py > "$O/probe.log" 2>&1; echo "probe rc=$?"; tail -14 "$O/probe.log"
    continue
  fi
  if [ "$m" = first ]; then   # the first encode of a process, many processes, fence behind every buffer
    for i in $(seq 1 ${FIRST_N:-20}); do
      MJH_GUARD=2 timeout 300 python tools/first_encode.py $i >> "$O/first.log" 2>&1 || echo "first-encode process $i failed rc=$?"
    done
    grep -c '"ok": true' "$O/first.log"; grep -i "fault\|error" "$O/first.log" | head
    continue
  fi
  MJH_GUARD=$m timeout 2400 python -m pytest tests -q -m gpu -n 4 -p no:cacheprovider "$@" > "$O/suite_mode$m.log" 2>&1
  echo "mode $m rc=$?"; tail -6 "$O/suite_mode$m.log"; grep -i "crashed\|fault\|MJH_GUARD" "$O/suite_mode$m.log" | head -20
done
# the BASELINE configurations at full size under the fence (bit-exact check against the reference included): GUARD_BENCH="metric c3 c5 c5t"
for c in $GUARD_BENCH; do
  MJH_GUARD=${GUARD_BENCH_MODE:-2} timeout 900 python bench.py --config $c --steps 3 --warmup 1 --no-cpu-baseline --no-inflight-leg --host-seconds 0.5 > "$O/bench_$c.log" 2>&1
  echo "bench $c under MJH_GUARD rc=$?"; python - "$O/bench_$c.log" <<'PY'
import json,sys
try:
    d=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1]); print('  ', d['config'].get('config_key'), d['ms_per_step'], d['bit_exact'])
except Exception as e:
    print('  ERR', e); print(open(sys.argv[1]).read()[-800:])
PY
done
