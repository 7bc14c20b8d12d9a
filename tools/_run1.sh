cd /tmp && export TMPDIR=/tmp; cd "$GRAFT_REPO_ROOT"
timeout 200 python tools/bench_variants.py --env MJH_FUSE --variants 1,0,3 > gpurun_out/var7.log 2>&1; cut -c1-420 gpurun_out/var7.log
timeout 400 python -m pytest tests -q -m gpu --timeout 120 -x > gpurun_out/t_all.log 2>&1; tail -6 gpurun_out/t_all.log
