#!/bin/bash
# SQ counter passes (each in its own rocprofv3 run with --kernel-trace only) over tools/bench_variants.py
# usage (on the GPU box): bash tools/pmc_passes.sh TAG [variant]
TAG=${1:-pmc}; VAR=${2:-0}
cd /tmp && export TMPDIR=/tmp
cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/$TAG; mkdir -p "$O"
CMD="python tools/bench_variants.py --variants $VAR --steps 2"
timeout 150 rocprofv3 --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INST_LEVEL_VMEM SQ_INSTS_SMEM SQ_INST_LEVEL_SMEM --kernel-trace -d "$O" -o a -- $CMD > "$O/a.log" 2>&1
timeout 150 rocprofv3 --pmc SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_ANY SQ_IFETCH SQ_IFETCH_LEVEL SQ_INST_LEVEL_LDS --kernel-trace -d "$O" -o b -- $CMD > "$O/b.log" 2>&1
timeout 150 rocprofv3 --pmc SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_THREAD_CYCLES_VALU SQ_INSTS_LDS SQ_INSTS_VALU SQ_INSTS_SALU SQ_INST_CYCLES_VMEM SQ_INSTS_BRANCH --kernel-trace -d "$O" -o c -- $CMD > "$O/c.log" 2>&1
timeout 150 rocprofv3 --pmc SQ_LEVEL_WAVES SQ_BUSY_CU_CYCLES GRBM_GUI_ACTIVE SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE --kernel-trace -d "$O" -o d -- $CMD > "$O/d.log" 2>&1
ls "$O"
python tools/pmc_sq.py $(find "$O" -name "*_results.db") --kernels=${3:-k_trellis_ac,k_dct_quant,k_enc_write,k_color,k_stats,k_enc_len} > "$O/sq.json"; cat "$O/sq.json"
