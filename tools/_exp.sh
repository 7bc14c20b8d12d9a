cd /tmp && export TMPDIR=/tmp; cd "$GRAFT_REPO_ROOT"
timeout 900 python -m pytest tests/test_gpu_fuzz.py tests/test_gpu_parity.py tests/test_gpu_host_path.py -q -m gpu --timeout 120 > gpurun_out/t12.log 2>&1; tail -12 gpurun_out/t12.log
