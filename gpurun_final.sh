cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
O=gpurun_out/r01e; mkdir -p $O
python bench.py > $O/bench_default.log 2>&1; tail -1 $O/bench_default.log | cut -c1-300
rocprofv3 --kernel-trace --stats -d $O -o stats -- python bench.py --steps 20 --warmup 3 --no-cpu-baseline --batch 16 > $O/stats.log 2>&1
rocprofv3 --pmc FETCH_SIZE --kernel-trace -d $O -o fetch -- python bench.py --steps 2 --warmup 1 --no-cpu-baseline --batch 16 > $O/fetch.log 2>&1
rocprofv3 --pmc WRITE_SIZE --kernel-trace -d $O -o write -- python bench.py --steps 2 --warmup 1 --no-cpu-baseline --batch 16 > $O/write.log 2>&1
rocprofv3 --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_SALU SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_BUSY_CYCLES --kernel-trace -d $O -o sq -- python bench.py --steps 2 --warmup 1 --no-cpu-baseline --batch 16 > $O/sq.log 2>&1
python tools/bench_configs.py > $O/configs.jsonl 2> $O/configs.err; wc -l $O/configs.jsonl
python tools/bench_host.py > $O/host.json 2> $O/host.err; cat $O/host.json
