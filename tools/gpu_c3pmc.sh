#!/bin/bash
# SQ counters of the C3 chain kernels (two passes, each its own rocprofv3 run with --kernel-trace only); usage: gpu_c3pmc.sh TAG
TAG=${1:-c3pmc}
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
O=gpurun_out/$TAG; mkdir -p "$O"
CMD="python bench.py --config c3 --steps 1 --warmup 1 --no-cpu-baseline --no-host-leg --no-inflight-leg"
timeout 200 rocprofv3 --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_ANY SQ_BUSY_CU_CYCLES --kernel-trace -d "$O" -o a -- $CMD > "$O/a.log" 2>&1
timeout 200 rocprofv3 --pmc SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_THREAD_CYCLES_VALU SQ_INSTS_LDS SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR --kernel-trace -d "$O" -o c -- $CMD > "$O/c.log" 2>&1
python tools/pmc_sq.py $(find "$O" -name "*_results.db") --kernels=k_pp_emit,k_pp_write,k_pp_len,k_pp_stats,k_prog_stuff,k_offsets,k_pp_chunk > "$O/sq.json"; cat "$O/sq.json"
