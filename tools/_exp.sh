cd /tmp && export TMPDIR=/tmp; cd "$GRAFT_REPO_ROOT"
for c in c5 c5t; do timeout 300 python tools/bench_variants.py --config $c --batch 1 --env MJH_NOP --variants 0 --steps 20 > gpurun_out/exp_$c.log 2>&1; tail -1 gpurun_out/exp_$c.log | cut -c1-400; done
timeout 300 python tools/bench_variants.py --config c3 --batch 32 --env MJH_NOP --variants 0 --steps 8 > gpurun_out/exp_c3h.log 2>&1; tail -1 gpurun_out/exp_c3h.log | cut -c1-330
timeout 600 python -m pytest tests/test_gpu_large.py tests/test_gpu_parity.py -q -m gpu --timeout 300 -x > gpurun_out/t14.log 2>&1; tail -4 gpurun_out/t14.log
