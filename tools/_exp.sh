cd /tmp && export TMPDIR=/tmp; cd "$GRAFT_REPO_ROOT"
O=gpurun_out/sqc3; mkdir -p "$O"
CMD="python tools/bench_variants.py --config c3 --batch 16 --env MJH_NOP --variants 0 --steps 1"
timeout 200 rocprofv3 --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INST_LEVEL_VMEM SQ_INSTS_SMEM SQ_INST_LEVEL_SMEM --kernel-trace -d "$O" -o a -- $CMD > "$O/a.log" 2>&1
timeout 200 rocprofv3 --pmc SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_ANY SQ_IFETCH SQ_IFETCH_LEVEL SQ_INST_LEVEL_LDS --kernel-trace -d "$O" -o b -- $CMD > "$O/b.log" 2>&1
timeout 200 rocprofv3 --pmc SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_THREAD_CYCLES_VALU SQ_INSTS_LDS SQ_INSTS_VALU SQ_INSTS_SALU SQ_INST_CYCLES_VMEM SQ_INSTS_BRANCH --kernel-trace -d "$O" -o c -- $CMD > "$O/c.log" 2>&1
python tools/pmc_sq.py $(find "$O" -name "*_results.db") --kernels=k_pp_write,k_pp_len,k_pp_stats > "$O/sq.json"; cat "$O/sq.json"
