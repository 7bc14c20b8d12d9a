#!/bin/bash
# SQ counter passes of the C3 chain and of the metric's kernels in one call (each pass its own rocprofv3 run, --kernel-trace only); usage: gpu_sq_all.sh TAG
TAG=${1:-r06sq}
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
O=gpurun_out/$TAG; mkdir -p "$O"
Q="--no-cpu-baseline --no-host-leg --no-inflight-leg --other-configs none --verify 1"
for cfg in c3 metric; do
  CMD="python bench.py --config $cfg --steps 2 --warmup 1 $Q"
  timeout 250 rocprofv3 --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_ANY SQ_BUSY_CU_CYCLES --kernel-trace -d "$O" -o ${cfg}_a -- $CMD > "$O/${cfg}_a.log" 2>&1
  timeout 250 rocprofv3 --pmc SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_THREAD_CYCLES_VALU SQ_INSTS_LDS SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR --kernel-trace -d "$O" -o ${cfg}_c -- $CMD > "$O/${cfg}_c.log" 2>&1
  timeout 250 rocprofv3 --kernel-trace --stats -d "$O" -o ${cfg}_stats -- python bench.py --config $cfg --steps 10 --warmup 3 $Q > "$O/${cfg}_stats.log" 2>&1
  python tools/pmc_sq.py $(find "$O" -name "${cfg}_[ac]_results.db") > "$O/${cfg}_sq.json"
  python tools/rocprof_summary.py "$(find "$O" -name "${cfg}_stats_results.db" | head -1)" > "$O/${cfg}_kernel_stats.csv"
done
ls -la "$O" | head -30
