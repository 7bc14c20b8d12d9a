#!/bin/bash
# per-kernel times of the C3 chain (rocprofv3 kernel trace of a short bench run); usage: gpu_c3prof.sh TAG [ENV=..]
cd "$GRAFT_REPO_ROOT" || exit 1
T=${1:-c3prof}; shift
O=gpurun_out/$T; mkdir -p "$O"; export TMPDIR=/tmp
env "$@" timeout 300 rocprofv3 --kernel-trace --stats -d "$O" -o c3 -f csv -- python bench.py --config c3 --steps 8 --warmup 2 --no-cpu-baseline --no-host-leg --no-inflight-leg > "$O/run.log" 2>&1
python - "$O" <<'PY'
import csv,sys,glob
f=glob.glob(sys.argv[1]+'/**/c3_kernel_stats.csv', recursive=True)
rows=list(csv.DictReader(open(f[0])))
for r in rows[:24]:
    print(f"{r['Name'][:70]:70s} {r['Calls']:>5s} {float(r['TotalDurationNs'])/1e3/int(r['Calls']):9.1f} us  {r['Percentage']}")
PY
