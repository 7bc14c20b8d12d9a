#!/bin/bash
# PC sampling of the metric workload (rocprofv3 beta feature; own run, no counters): WHERE inside k_trellis_ac_v3 / k_trellis_dc3 /
# k_dct_quant the issue cycles go -- per instruction, with the exec mask of every sample (lane activity per instruction) and,
# with the stochastic method, the reason a wave did not issue.  Round 4 could only count instructions per kernel (SQ_INSTS_VALU).
# usage: gpurun --timeout 600 -- 'bash tools/gpu_pcsample.sh [stochastic|host_trap]'
cd "$GRAFT_REPO_ROOT" || exit 1
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
M=${1:-stochastic}; O=gpurun_out/pcs_$M; mkdir -p "$O"
if [ "$M" = stochastic ]; then U="--pc-sampling-unit cycles --pc-sampling-interval 1048576"; else U="--pc-sampling-unit time --pc-sampling-interval 100"; fi
timeout 300 rocprofv3 --kernel-trace --pc-sampling-beta-enabled --pc-sampling-method $M $U --output-format csv -d "$O" -o pcs -- \
  python tools/bench_variants.py --variants 0 --env MJH_SORTED_UQ --steps 4 > "$O/run.log" 2>&1
echo "rc $?"; tail -3 "$O/run.log"; ls -la "$O" | head
f=$(ls "$O"/*pc_sampling*.csv 2>/dev/null | head -1)
[ -n "$f" ] && python tools/pcsample_summary.py "$f" > "$O/summary.txt" && head -60 "$O/summary.txt"
# keep the upload small: the raw sample file can be hundreds of MB
[ -n "$f" ] && gzip -f "$f" && ls -la "$O"
