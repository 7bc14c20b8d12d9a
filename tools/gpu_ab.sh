#!/bin/bash
# one GPU call: A/B of an environment knob on the metric workload (files must be identical), then optional parity tests
# usage: bash tools/gpu_ab.sh TAG ENVNAME VARIANTS [pytest args...]
cd "$GRAFT_REPO_ROOT" || exit 1
TAG=$1; ENVN=$2; VARS=$3; shift 3
O=gpurun_out/$TAG; mkdir -p "$O"
[ -x tools/probes/dpp_probe ] && tools/probes/dpp_probe > "$O/dpp_probe.txt" 2>&1
timeout 400 python tools/bench_variants.py --env "$ENVN" --variants "$VARS" --steps 10 > "$O/variants.log" 2>&1
grep '^{' "$O/variants.log" | cut -c1-520
if [ $# -gt 0 ]; then timeout 1200 python -m pytest "$@" -x -q > "$O/pytest.log" 2>&1; tail -4 "$O/pytest.log"; fi
