cd /tmp && export TMPDIR=/tmp; cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r02e; mkdir -p $O
timeout 150 rocprofv3 --kernel-trace --stats -d "$O" -o c3_stats -- python bench.py --config c3 --steps 10 --warmup 3 --no-cpu-baseline --no-host-leg --no-inflight-leg --verify 1 > "$O/c3_stats.log" 2>&1
tail -1 $O/c3_stats.log | cut -c1-200
