cd /tmp && export TMPDIR=/tmp; cd "$GRAFT_REPO_ROOT"
timeout 900 python -m pytest tests -q -m gpu --timeout 120 > gpurun_out/t_all.log 2>&1; tail -15 gpurun_out/t_all.log
