cd /tmp && export TMPDIR=/tmp; cd "$GRAFT_REPO_ROOT"
timeout 400 python bench.py > gpurun_out/bench_final.log 2>&1; tail -1 gpurun_out/bench_final.log | python -c "
import sys, json
j = json.loads(sys.stdin.read()); print(j['value'], j['ms_per_step'], j['steps'], j.get('pipelined'), j['roofline']['frac'], j['host_inclusive'].get('value'), j['cpu_baseline'].get('value'), j['bit_exact']['ok'])"
timeout 200 python bench.py --config c4 --steps 5 --no-cpu-baseline --no-host-leg --verify 2 > gpurun_out/bench_c4_final.log 2>&1; tail -1 gpurun_out/bench_c4_final.log | cut -c1-100; tail -1 gpurun_out/bench_c4_final.log | python -c "
import sys, json
j = json.loads(sys.stdin.read()); print(j.get('pipelined'))"
