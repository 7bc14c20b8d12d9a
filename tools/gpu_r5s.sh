#!/bin/bash
# Round 5: the AC trellis' build phase with its row constants fetched eight positions at a time (on top of the same change in
# k_dct_quant): library of the tree against the k_dct_quant-only library and the library of commit d05e816, alternating
cd "$GRAFT_REPO_ROOT" || exit 1
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r5s; mkdir -p "$O"
Q="--no-cpu-baseline --no-host-leg --no-inflight-leg --verify 2 --other-configs none"
line() { python -c "import sys,json; d=json.loads(sys.stdin.read()); r=d['roofline']['kernel_ms_per_call(untimed pass, every kernel bracketed)']; print(d['value'], d['ms_per_step'], d['bit_exact']['ok'] if isinstance(d.get('bit_exact'),dict) else d.get('bit_exact'), {k: r[k] for k in list(r)[:5]})"; }
for v in k2only new d05e816 k2only new; do
  lib=""; [ $v != new ] && lib=$PWD/gpurun_ab/libmozjpeg_hip_$v.so
  MOZJPEG_AMD_LIB=$lib timeout 200 python bench.py --steps 150 --warmup 30 $Q > "$O/metric_$v.log" 2>&1
  echo "metric $v $(tail -1 "$O/metric_$v.log" | line)"
done
for c in c5t c3 c2; do for v in k2only new; do
  lib=""; [ $v != new ] && lib=$PWD/gpurun_ab/libmozjpeg_hip_$v.so
  MOZJPEG_AMD_LIB=$lib timeout 200 python bench.py --config $c $Q > "$O/${c}_$v.log" 2>&1
  echo "$c $v $(tail -1 "$O/${c}_$v.log" | line)"
done; done
