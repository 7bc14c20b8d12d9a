#!/bin/bash
# what the driver does at round end, on a fresh box: smoke(), the default bench line; plus the arithmetic tests under both
# memory fences (MJH_GUARD=2 / 3) for the kernels that changed last
cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/final; mkdir -p "$O"
timeout 600 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" > "$O/smoke.log" 2>&1; echo "smoke rc=$?"; tail -2 "$O/smoke.log"
timeout 900 python bench.py > "$O/bench_default.json" 2> "$O/bench_default.err"; echo "bench rc=$?"
python - "$O/bench_default.json" <<'PY'
import json,sys
d=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1]); r=d['roofline']
print(d['value'], d['ms_per_step'], d['bit_exact_vs_reference'], 'frac', r['frac'], 'traffic', r['traffic'], str(r['traffic_source'])[:120])
print('host', d.get('value_host_inclusive'), 'cpu', d['cpu_baseline']['value'])
PY
for g in 2 3; do
  MJH_GUARD=$g timeout 600 python -m pytest tests -q -m gpu -k "arith" -n 4 -p no:cacheprovider 2>&1 | tail -2
done
