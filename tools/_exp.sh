cd /tmp && export TMPDIR=/tmp; cd "$GRAFT_REPO_ROOT"
timeout 1200 python -m pytest tests -q -m gpu --timeout 300 > gpurun_out/t_final.log 2>&1; tail -4 gpurun_out/t_final.log
timeout 300 python tools/bench_variants.py --config c3 --batch 64 --env MJH_NOP --variants 0 --steps 5 > gpurun_out/exp_c3_64.log 2>&1; tail -1 gpurun_out/exp_c3_64.log | cut -c1-120
timeout 300 python tools/bench_variants.py --config metric --batch 128 --env MJH_NOP --variants 0 --steps 10 > gpurun_out/exp_m128.log 2>&1; tail -1 gpurun_out/exp_m128.log | cut -c1-120
timeout 300 python bench.py --config c3 --no-cpu-baseline --no-host-leg > gpurun_out/c3_final.log 2>&1; tail -1 gpurun_out/c3_final.log | cut -c1-300
