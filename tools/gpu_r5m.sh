#!/bin/bash
# Round 5, final profile set: tools/profile_round.sh r05m 64 all (bench lines of every configuration, kernel traces, FETCH / WRITE
# passes, two SQ passes, the drop-in throughput), after the parity cases of the last host-side changes on the chip
cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/r05m; mkdir -p "$O"
timeout 400 python -m pytest tests/test_gpu_parity.py tests/test_gpu_dropin.py -q -m gpu -k "script or q_opt or samp or scan_scripts" > "$O/last_cases.log" 2>&1; tail -2 "$O/last_cases.log"
bash tools/profile_round.sh r05m 64 all
MOZJPEG_HIP_TIMING=1 LD_LIBRARY_PATH=$PWD/oracle/_ref:$LD_LIBRARY_PATH LD_PRELOAD=$PWD/mozjpeg_amd/libmozjpeg_hip_jpeg62.so timeout 120 tests/native/mt_bench 1 1200 3840 2160 75 baseline > "$O/mt1.json" 2> "$O/mt1.err"; tail -1 "$O/mt1.json" | cut -c1-200; grep timing "$O/mt1.err"
