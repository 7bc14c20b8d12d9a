#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/${1:-steps}; mkdir -p "$O"
Q="--no-cpu-baseline --no-host-leg --no-inflight-leg --verify 1"
for cfg in "20 5" "100 10" "500 50"; do set -- $cfg
  timeout 300 python bench.py --steps $1 --warmup $2 $Q > "$O/b_$1.log" 2>&1; python - "$O/b_$1.log" <<'PY'
import json,sys
d=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1]); print(sys.argv[1], d['steps'], d['ms_per_step'], d['value'], d['roofline']['kernel'], d['roofline']['kernel_ms'])
PY
done
MJH_SPLIT=1 timeout 300 python bench.py --steps 20 --warmup 5 $Q > "$O/b_split1.log" 2>&1; tail -1 "$O/b_split1.log" | cut -c1-200
MJH_SPLIT=1 timeout 300 python bench.py --steps 500 --warmup 50 $Q > "$O/b_split1_500.log" 2>&1; tail -1 "$O/b_split1_500.log" | cut -c1-200
rocm-smi --showclocks --showpower 2>/dev/null | head -30 > "$O/smi.txt"
