cd /tmp && export TMPDIR=/tmp; cd "$GRAFT_REPO_ROOT"
for c in c3 c5; do timeout 200 python bench.py --config $c --steps 10 --no-cpu-baseline --no-host-leg --verify 1 > gpurun_out/bench_${c}_fin.log 2>&1; tail -1 gpurun_out/bench_${c}_fin.log | python -c "
import sys, json
j = json.loads(sys.stdin.read()); print(j['config']['config_key'], j['value'], j['ms_per_step'], j.get('pipelined'), j['bit_exact']['ok'])"; done
timeout 100 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -1
