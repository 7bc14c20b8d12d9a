cd /tmp && export TMPDIR=/tmp; cd "$GRAFT_REPO_ROOT"
timeout 300 python tools/bench_variants.py --config metric --batch 64 --env MJH_TILE --variants 0,1 --steps 10 > gpurun_out/exp_tile.log 2>&1; tail -2 gpurun_out/exp_tile.log | cut -c1-300
