cd /tmp && export TMPDIR=/tmp; cd "$GRAFT_REPO_ROOT"
timeout 600 python -m pytest tests/test_gpu_parity.py tests/test_gpu_host_path.py -q -m gpu --timeout 120 -x > gpurun_out/t11.log 2>&1; tail -6 gpurun_out/t11.log
timeout 300 python tools/bench_variants.py --config c3 --batch 32 --env MJH_COMPACT --variants 0,1 --steps 5 > gpurun_out/exp_c3c.log 2>&1; tail -2 gpurun_out/exp_c3c.log | cut -c1-520
