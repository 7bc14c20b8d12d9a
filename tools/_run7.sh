cd /tmp && export TMPDIR=/tmp; cd "$GRAFT_REPO_ROOT"
timeout 900 python -m pytest tests/test_gpu_dropin.py tests/test_standalone_api.py -q -m gpu --timeout 300 > gpurun_out/t7.log 2>&1; tail -15 gpurun_out/t7.log
