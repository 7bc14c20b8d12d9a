cd /tmp && export TMPDIR=/tmp; cd "$GRAFT_REPO_ROOT"
timeout 1200 python -m pytest tests -q -m gpu --timeout 300 > gpurun_out/t_all.log 2>&1; tail -8 gpurun_out/t_all.log
bash tools/profile_round.sh r02c 64 all
