cd /tmp && export TMPDIR=/tmp; cd "$GRAFT_REPO_ROOT"
timeout 150 python tools/bench_variants.py --variants 0,1,2 > gpurun_out/var4.log 2>&1; cut -c1-330 gpurun_out/var4.log
timeout 200 python -m pytest tests/test_gpu_parity.py -q --timeout 120 > gpurun_out/t_par.log 2>&1; tail -4 gpurun_out/t_par.log
O=gpurun_out/st_r02c; mkdir -p $O
timeout 150 rocprofv3 --kernel-trace --stats -d $O -o st -- python tools/bench_variants.py --variants 0 --steps 5 > $O/st.log 2>&1
python tools/rocprof_summary.py $O/st_kernel_stats.csv 2>/dev/null | head -12 || head -12 $O/st_kernel_stats.csv
ls $O
