#!/bin/bash
# Round 5, final tree: the whole GPU suite, the profile set r05t (kernel trace, traffic and SQ passes of the metric; bench lines,
# traffic passes and kernel traces of the other configurations), then -- with the traffic summaries of THIS tree in place -- the
# default bench line.  tools/collect_profiles.sh turns gpurun_out/r05t into profiles/r05t_* (here for the bench line's
# traffic look-up, and again in the build container, where git knows the commit).
cd "$GRAFT_REPO_ROOT" || exit 1
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
TAG=r05t; BATCH=64
O=gpurun_out/$TAG; mkdir -p "$O"
echo "== suite"; timeout 900 python -m pytest tests -q -m gpu > "$O/suite.log" 2>&1; tail -3 "$O/suite.log"
Q="--no-cpu-baseline --no-host-leg --no-inflight-leg --verify 1 --other-configs none"
timeout 200 rocprofv3 --kernel-trace --stats -d "$O" -o stats -- python bench.py --steps 20 --warmup 3 $Q --batch $BATCH > "$O/stats.log" 2>&1
timeout 200 rocprofv3 --pmc FETCH_SIZE --kernel-trace -d "$O" -o fetch -- python bench.py --steps 2 --warmup 1 $Q --batch $BATCH > "$O/fetch.log" 2>&1
timeout 200 rocprofv3 --pmc WRITE_SIZE --kernel-trace -d "$O" -o write -- python bench.py --steps 2 --warmup 1 $Q --batch $BATCH > "$O/write.log" 2>&1
timeout 200 rocprofv3 --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_SALU SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_BUSY_CYCLES --kernel-trace -d "$O" -o sq -- python bench.py --steps 2 --warmup 1 $Q --batch $BATCH > "$O/sq.log" 2>&1
timeout 200 rocprofv3 --pmc SQ_THREAD_CYCLES_VALU SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_ANY SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR GRBM_GUI_ACTIVE --kernel-trace -d "$O" -o sq2 -- python bench.py --steps 2 --warmup 1 $Q --batch $BATCH > "$O/sq2.log" 2>&1
echo "== passes of the metric done"
for c in c2 c5 c5t c3; do
  timeout 200 rocprofv3 --pmc FETCH_SIZE --kernel-trace -d "$O" -o ${c}_fetch -- python bench.py --config $c --steps 2 --warmup 1 $Q > "$O/${c}_fetch.log" 2>&1
  timeout 200 rocprofv3 --pmc WRITE_SIZE --kernel-trace -d "$O" -o ${c}_write -- python bench.py --config $c --steps 2 --warmup 1 $Q > "$O/${c}_write.log" 2>&1
done
timeout 200 rocprofv3 --kernel-trace --stats -d "$O" -o c3_stats -- python bench.py --config c3 --steps 10 --warmup 3 $Q > "$O/c3_stats.log" 2>&1
timeout 200 rocprofv3 --kernel-trace --stats -d "$O" -o c5_stats -- python bench.py --config c5 --steps 5 --warmup 2 $Q > "$O/c5_stats.log" 2>&1
timeout 200 rocprofv3 --kernel-trace --stats -d "$O" -o c5t_stats -- python bench.py --config c5t --steps 5 --warmup 2 $Q > "$O/c5t_stats.log" 2>&1
# the summaries of this tree's passes, so that the lines below quote THEIR traffic
: > "$O/bench_default.log"; bash tools/collect_profiles.sh $TAG $BATCH > "$O/collect.log" 2>&1
echo "== config lines"
for c in c2 c3 c4 c5 c5t; do
  timeout 300 python bench.py --config $c --cpu-budget 5 --no-host-leg > "$O/bench_$c.log" 2>&1; tail -1 "$O/bench_$c.log" | cut -c90-230
done
echo "== bench"; timeout 400 python bench.py > "$O/bench_default.log" 2>&1; tail -1 "$O/bench_default.log" | cut -c1-330
