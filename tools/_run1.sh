cd /tmp && export TMPDIR=/tmp; cd "$GRAFT_REPO_ROOT"
timeout 400 python -m pytest tests/test_gpu_parity.py -q -m gpu --timeout 120 -x > gpurun_out/t_par.log 2>&1; tail -12 gpurun_out/t_par.log
timeout 200 python tools/bench_variants.py --config c3 --batch 8 --env MJH_NOP --variants 0 --steps 10 > gpurun_out/c3a.log 2>&1; tail -1 gpurun_out/c3a.log | cut -c1-900
O=gpurun_out/st_c3c; mkdir -p $O
timeout 200 rocprofv3 --kernel-trace --stats -d $O -o st -- python tools/bench_variants.py --config c3 --batch 8 --env MJH_NOP --variants 0 --steps 5 > $O/st.log 2>&1
python tools/rocprof_summary.py $O/st_results.db | head -24
