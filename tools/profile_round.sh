#!/bin/bash
# How profiles/rNN_* are produced (run on the GPU box: /usr/local/graft/bin/gpurun -- 'bash tools/profile_round.sh r02x [64] [all]'):
# the bench line, the rocprofv3 kernel-trace summary of the same command, and the PMC passes -- each counter set in
# its own run with --kernel-trace only, as /opt/skills/guides/MI355X_MICROARCH.md prescribes.  Results land in
# gpurun_out/<tag>/ and are turned into profiles/<tag>_* by tools/rocprof_summary.py and tools/pmc_traffic.py
# (tools/collect_profiles.sh).  Every command runs under its own timeout.
TAG=${1:-r02x}
BATCH=${2:-64}
cd /tmp && export TMPDIR=/tmp
cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/$TAG; mkdir -p "$O"
Q="--no-cpu-baseline --no-host-leg --no-inflight-leg --verify 1"
timeout 400 python bench.py > "$O/bench_default.log" 2>&1; tail -1 "$O/bench_default.log" | cut -c1-400
timeout 200 rocprofv3 --kernel-trace --stats -d "$O" -o stats -- python bench.py --steps 20 --warmup 3 $Q --batch "$BATCH" > "$O/stats.log" 2>&1
timeout 200 rocprofv3 --pmc FETCH_SIZE --kernel-trace -d "$O" -o fetch -- python bench.py --steps 2 --warmup 1 $Q --batch "$BATCH" > "$O/fetch.log" 2>&1
timeout 200 rocprofv3 --pmc WRITE_SIZE --kernel-trace -d "$O" -o write -- python bench.py --steps 2 --warmup 1 $Q --batch "$BATCH" > "$O/write.log" 2>&1
timeout 200 rocprofv3 --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_SALU SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_BUSY_CYCLES --kernel-trace -d "$O" -o sq -- python bench.py --steps 2 --warmup 1 $Q --batch "$BATCH" > "$O/sq.log" 2>&1
# lane activity (SQ_THREAD_CYCLES_VALU / SQ_INSTS_VALU = active lanes per VALU instruction), LDS conflicts, memory instructions
timeout 200 rocprofv3 --pmc SQ_THREAD_CYCLES_VALU SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_ANY SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR GRBM_GUI_ACTIVE --kernel-trace -d "$O" -o sq2 -- python bench.py --steps 2 --warmup 1 $Q --batch "$BATCH" > "$O/sq2.log" 2>&1
# configuration 3 (progressive + scan search): bench line, kernel trace, traffic
timeout 300 python bench.py --config c3 --cpu-budget 10 --no-host-leg > "$O/bench_c3.log" 2>&1; tail -1 "$O/bench_c3.log" | cut -c1-300
timeout 200 rocprofv3 --kernel-trace --stats -d "$O" -o c3_stats -- python bench.py --config c3 --steps 10 --warmup 3 $Q > "$O/c3_stats.log" 2>&1
timeout 200 rocprofv3 --pmc FETCH_SIZE --kernel-trace -d "$O" -o c3_fetch -- python bench.py --config c3 --steps 2 --warmup 1 $Q > "$O/c3_fetch.log" 2>&1
timeout 200 rocprofv3 --pmc WRITE_SIZE --kernel-trace -d "$O" -o c3_write -- python bench.py --config c3 --steps 2 --warmup 1 $Q > "$O/c3_write.log" 2>&1
if [ "$3" = "all" ]; then
  for c in c2 c4 c5 c5t; do
    timeout 400 python bench.py --config $c --cpu-budget 10 --no-host-leg > "$O/bench_$c.log" 2>&1; tail -1 "$O/bench_$c.log" | cut -c1-300
  done
  # traffic passes of the other configurations (bench.py's roofline.traffic is looked up per configuration)
  for c in c2 c4 c5 c5t; do
    timeout 200 rocprofv3 --pmc FETCH_SIZE --kernel-trace -d "$O" -o ${c}_fetch -- python bench.py --config $c --steps 2 --warmup 1 $Q > "$O/${c}_fetch.log" 2>&1
    timeout 200 rocprofv3 --pmc WRITE_SIZE --kernel-trace -d "$O" -o ${c}_write -- python bench.py --config $c --steps 2 --warmup 1 $Q > "$O/${c}_write.log" 2>&1
  done
  timeout 200 rocprofv3 --kernel-trace --stats -d "$O" -o c5_stats -- python bench.py --config c5 --steps 5 --warmup 2 $Q > "$O/c5_stats.log" 2>&1
  timeout 200 rocprofv3 --kernel-trace --stats -d "$O" -o c5t_stats -- python bench.py --config c5t --steps 5 --warmup 2 $Q > "$O/c5t_stats.log" 2>&1
  timeout 600 python tools/bench_dropin.py > "$O/dropin.json" 2> "$O/dropin.err"; tail -5 "$O/dropin.err"
fi
