#!/bin/bash
# one GPU call: the AC trellis over image ranges (MJH_TRELLIS_CHUNKS) on the metric workload and the other configurations,
# live trellis interval with two batches in flight, then the full-size parity tests under the chosen setting
# usage: bash tools/gpu_chunks.sh TAG VARIANTS [ENV=VAL for the pytest leg]
cd "$GRAFT_REPO_ROOT" || exit 1
TAG=$1; VARS=$2; shift 2
O=gpurun_out/$TAG; mkdir -p "$O"
for cfg in metric c2 c3; do
  timeout 500 python tools/bench_variants.py --config $cfg --batch $([ $cfg = c3 ] && echo 32 || echo 64) --env MJH_TRELLIS_CHUNKS --variants "$VARS" --steps 60 --prof 2 > "$O/variants_$cfg.log" 2>&1
  echo "== $cfg"; grep '^{' "$O/variants_$cfg.log" | cut -c1-330
done
if [ $# -gt 0 ]; then env "$@" timeout 1500 python -m pytest tests/test_gpu_large.py tests/test_gpu_parity.py -x -q > "$O/pytest.log" 2>&1; tail -4 "$O/pytest.log"; fi
