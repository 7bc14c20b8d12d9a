cd /tmp && export TMPDIR=/tmp; cd "$GRAFT_REPO_ROOT"
timeout 600 python -m pytest tests/test_gpu_parity.py tests/test_gpu_fuzz.py tests/test_gpu_host_path.py -q -m gpu --timeout 300 -x > gpurun_out/t16.log 2>&1; tail -4 gpurun_out/t16.log
timeout 300 python tools/bench_variants.py --config c3 --batch 32 --env MJH_NOP --variants 0 --steps 10 > gpurun_out/exp_c3s.log 2>&1; tail -1 gpurun_out/exp_c3s.log | cut -c1-480
