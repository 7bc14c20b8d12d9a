#!/bin/bash
# parity tests of the trellis paths + the metric line with its per-kernel table + single-frame timing
cd /tmp && export TMPDIR=/tmp
cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/${1:-quick}; mkdir -p "$O"
timeout 1200 python -m pytest tests/test_gpu_parity.py tests/test_gpu_fuzz.py -q -m gpu -n 4 -p no:cacheprovider -x 2>&1 | tail -3
Q="--no-cpu-baseline --no-host-leg --no-inflight-leg --verify 4"
timeout 300 rocprofv3 --kernel-trace --stats -d "$O" -o st -- python bench.py --steps 20 --warmup 3 $Q > "$O/stats.log" 2>&1
python tools/rocprof_summary.py "$(find $O -name 'st_results.db' | head -1)" | head -8
timeout 300 python bench.py --steps 200 $Q > "$O/bench.log" 2>&1
python - "$O/bench.log" <<'PY'
import json,sys
d=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1]); r=d['roofline']
print(d['ms_per_step'], d['value'], d['bit_exact']['ok'], r['kernel_ms'])
PY
bash tools/gpu_single.sh ${1:-quick}_single 2>&1 | head -3
