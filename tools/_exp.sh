cd /tmp && export TMPDIR=/tmp; cd "$GRAFT_REPO_ROOT"
timeout 600 python -m pytest tests/test_gpu_parity.py tests/test_gpu_host_path.py -q -m gpu --timeout 120 -x > gpurun_out/t8.log 2>&1; tail -4 gpurun_out/t8.log
timeout 300 python bench.py --config c3 --no-cpu-baseline --no-host-leg --steps 10 > gpurun_out/c3_32.log 2>&1; tail -1 gpurun_out/c3_32.log | cut -c1-420
timeout 300 python bench.py --config c3 --batch 8 --no-cpu-baseline --no-host-leg --steps 10 --verify 1 > gpurun_out/c3_8.log 2>&1; tail -1 gpurun_out/c3_8.log | cut -c1-300
