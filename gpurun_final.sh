cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
O=gpurun_out/r01f; mkdir -p $O
python bench.py > $O/bench_default.log 2>&1; tail -1 $O/bench_default.log | cut -c1-300
rocprofv3 --kernel-trace --stats -d $O -o stats -- python bench.py --steps 20 --warmup 3 --no-cpu-baseline --batch 64 > $O/stats.log 2>&1
rocprofv3 --pmc FETCH_SIZE --kernel-trace -d $O -o fetch -- python bench.py --steps 2 --warmup 1 --no-cpu-baseline --batch 64 > $O/fetch.log 2>&1
rocprofv3 --pmc WRITE_SIZE --kernel-trace -d $O -o write -- python bench.py --steps 2 --warmup 1 --no-cpu-baseline --batch 64 > $O/write.log 2>&1
rocprofv3 --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_SALU SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_BUSY_CYCLES --kernel-trace -d $O -o sq -- python bench.py --steps 2 --warmup 1 --no-cpu-baseline --batch 64 > $O/sq.log 2>&1
echo skip configs
echo skip host
