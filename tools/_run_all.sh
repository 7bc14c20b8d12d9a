cd /tmp && export TMPDIR=/tmp; cd "$GRAFT_REPO_ROOT"
timeout 1200 python -m pytest tests -q -m gpu --timeout 300 > gpurun_out/t_all.log 2>&1; tail -5 gpurun_out/t_all.log
bash tools/profile_round.sh r02d 64 all
MJH_BENCH_DIST_BACKEND=gloo timeout 300 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29533 bench.py --gpus 2 --steps 20 --warmup 3 --verify 2 > gpurun_out/r02d/bench_2ranks_gloo.log 2>&1; tail -1 gpurun_out/r02d/bench_2ranks_gloo.log | cut -c1-300
