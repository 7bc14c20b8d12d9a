#!/bin/bash
# (the A/B switch MJH_PACK_MAIN used below is gone: profiles/r05n_dropin_handover_ab.md)
# Round 5: after the C5 fixes (k_stats_dc_mcu atomics, FDCT loop only with fused statistics) and the single-image hand-over on the
# main stream: C5 / metric timings, drop-in A/B (MJH_PACK_MAIN), the whole suite, C5's traffic passes
cd "$GRAFT_REPO_ROOT" || exit 1
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r5n; mkdir -p "$O"
Q="--no-cpu-baseline --no-host-leg --no-inflight-leg --verify 1"
for c in c5 c5t c2; do timeout 300 python bench.py --config $c --cpu-budget 5 --no-host-leg > "$O/bench_$c.log" 2>&1; tail -1 "$O/bench_$c.log" | cut -c90-260; done
timeout 200 rocprofv3 --kernel-trace --stats -d "$O" -o c5_stats -- python bench.py --config c5 --steps 5 --warmup 2 $Q > "$O/c5_stats.log" 2>&1
f=$(find "$O" -name "c5_stats_results.db" | head -1); [ -n "$f" ] && python tools/rocprof_summary.py "$f" | head -9
export LD_LIBRARY_PATH=$PWD/oracle/_ref:$LD_LIBRARY_PATH
for pm in 0 1 0 1; do
  MJH_PACK_MAIN=$pm MOZJPEG_HIP_TIMING=1 LD_PRELOAD=$PWD/mozjpeg_amd/libmozjpeg_hip_jpeg62.so timeout 120 tests/native/mt_bench 1 1500 3840 2160 75 baseline > "$O/mt_$pm.json" 2> "$O/mt_$pm.err"
  echo "MJH_PACK_MAIN=$pm $(tail -1 "$O/mt_$pm.json" | cut -c1-150)"; grep timing "$O/mt_$pm.err"
done
echo "== suite"; timeout 800 python -m pytest tests -q -m gpu > "$O/suite.log" 2>&1; tail -3 "$O/suite.log"
timeout 200 rocprofv3 --pmc FETCH_SIZE --kernel-trace -d "$O" -o c5_fetch -- python bench.py --config c5 --steps 2 --warmup 1 $Q > "$O/c5_fetch.log" 2>&1
timeout 200 rocprofv3 --pmc WRITE_SIZE --kernel-trace -d "$O" -o c5_write -- python bench.py --config c5 --steps 2 --warmup 1 $Q > "$O/c5_write.log" 2>&1
echo "== bench"; timeout 400 python bench.py > "$O/bench_default.log" 2>&1; tail -1 "$O/bench_default.log" | cut -c1-330
