cd /tmp && export TMPDIR=/tmp; cd "$GRAFT_REPO_ROOT"
timeout 1200 python -m pytest tests -q -m gpu --timeout 300 > gpurun_out/t6.log 2>&1; tail -25 gpurun_out/t6.log
timeout 300 python bench.py --config c5t --no-cpu-baseline --no-host-leg --steps 10 > gpurun_out/c5t.log 2>&1; tail -1 gpurun_out/c5t.log | cut -c1-250
