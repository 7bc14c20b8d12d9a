#!/bin/bash
# SQ counter passes of the metric workload (instructions, lanes per instruction, waits, LDS) -- the two --pmc sets of
# tools/profile_round.sh on their own.  usage: gpurun --timeout 500 -- 'bash tools/gpu_sq.sh TAG [bench args]'
TAG=${1:-sq}; shift
cd /tmp && export TMPDIR=/tmp
cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/$TAG; mkdir -p "$O"
Q="--no-cpu-baseline --no-host-leg --no-inflight-leg --verify 1 --other-configs none"
timeout 200 rocprofv3 --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_SALU SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_BUSY_CYCLES --kernel-trace -d "$O" -o sq -- python bench.py --steps 2 --warmup 1 $Q "$@" > "$O/sq.log" 2>&1
timeout 200 rocprofv3 --pmc SQ_THREAD_CYCLES_VALU SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_ANY SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR GRBM_GUI_ACTIVE --kernel-trace -d "$O" -o sq2 -- python bench.py --steps 2 --warmup 1 $Q "$@" > "$O/sq2.log" 2>&1
a=$(find "$O" -name "sq_results.db" | head -1); b=$(find "$O" -name "sq2_results.db" | head -1)
python tools/pmc_sq.py "$a" $b > "$O/pmc_sq.json" 2> "$O/pmc_sq.err"; python - <<PY
import json
d=json.load(open("$O/pmc_sq.json"))
for k,v in sorted(d.items(), key=lambda kv: -kv[1].get("SQ_INSTS_VALU",0))[:10]:
    iv=v.get("SQ_INSTS_VALU",0); 
    print("%-52s VALU %.4g  lanes/instr %.1f  LDS %.3g  conflicts/LDSinstr %.2f  wave_cycles %.3g  wait_inst_any %.3g  busy %.3g" % (k[:52], iv, v.get("SQ_THREAD_CYCLES_VALU",0)/max(iv,1), v.get("SQ_INSTS_LDS",0), v.get("SQ_LDS_BANK_CONFLICT",0)/max(v.get("SQ_INSTS_LDS",1),1), v.get("SQ_WAVE_CYCLES",0), v.get("SQ_WAIT_INST_ANY",0), v.get("SQ_BUSY_CYCLES",0)))
PY
