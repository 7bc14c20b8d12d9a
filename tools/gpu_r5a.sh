#!/bin/bash
# Round 5, first GPU call: the opt-in kernels of round 4 (tile-sorted planes, queue records, skiplow) on the chip for the
# first time -- their own tests, then A/B on the metric / C3 / C5t workloads -- then the whole suite on the default path
# (new: k_stats_dc_mcu, the q_opt reset in front of the FDCT), the bench line and a kernel trace of it.
# usage: gpurun --timeout 1100 -- 'bash tools/gpu_r5a.sh'
cd "$GRAFT_REPO_ROOT" || exit 1
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r5a; mkdir -p "$O"
echo "== 1. opt-in kernels: own tests"; MJH_TEST_SORTED=1 timeout 300 python -m pytest tests/test_gpu_parity.py -q -m gpu -k "tile_sorted or queue_records or skiplow" > "$O/optin_own.log" 2>&1; tail -3 "$O/optin_own.log"
ab() { # tag, config, env, variants, steps, [extra env assignment]
  timeout 300 env $6 python tools/bench_variants.py --config $2 --env $3 --variants $4 --steps $5 > "$O/ab_$1.log" 2>&1; echo "-- $1"; grep '^{' "$O/ab_$1.log" | cut -c1-560; grep -i "error\|fault\|Traceback" "$O/ab_$1.log" | head -3; }
echo "== 2. A/B"
ab rec metric MJH_TRELLIS_REC 0,1,0,1 10
ab sorted metric MJH_SORTED_UQ 0,1 10
ab sorted_tile metric MJH_SORTED_TILE 128,512 10 MJH_SORTED_UQ=1
ab rec_c3 c3 MJH_TRELLIS_REC 0,1 5
ab rec_c5t c5t MJH_TRELLIS_REC 0,1 5
ab skiplow_c3 c3 MJH_PP_SKIPLOW 0,1 5
echo "== 3. the whole suite, default path"; timeout 500 python -m pytest tests -q -m gpu -x > "$O/suite.log" 2>&1; tail -3 "$O/suite.log"
echo "== 4. bench"; timeout 300 python bench.py > "$O/bench.log" 2>&1; tail -1 "$O/bench.log" | cut -c1-900
echo "== 5. kernel trace"; timeout 200 rocprofv3 --kernel-trace --stats -d "$O" -o stats -- python bench.py --steps 20 --warmup 3 --no-cpu-baseline --no-host-leg --no-inflight-leg --verify 1 > "$O/stats.log" 2>&1
f=$(find "$O" -name "stats_results.db" | head -1); [ -n "$f" ] && python tools/rocprof_summary.py "$f" > "$O/kernel_stats.csv" && head -30 "$O/kernel_stats.csv" | cut -c1-220
