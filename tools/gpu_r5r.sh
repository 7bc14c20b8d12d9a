#!/bin/bash
# Round 5: k_dct_quant with its quantizer constants fetched eight positions at a time (one scalar round trip per 8 positions
# instead of two per position) against the library of commit d05e816, alternating on one box
cd "$GRAFT_REPO_ROOT" || exit 1
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r5r; mkdir -p "$O"
OLD=$PWD/gpurun_ab/libmozjpeg_hip_d05e816.so
Q="--no-cpu-baseline --no-host-leg --no-inflight-leg --verify 2 --other-configs none"
line() { python -c "import sys,json; d=json.loads(sys.stdin.read()); r=d['roofline']['kernel_ms_per_call(untimed pass, every kernel bracketed)']; print(d['value'], d['ms_per_step'], d['bit_exact']['ok'] if isinstance(d.get('bit_exact'),dict) else d.get('bit_exact'), {k: r[k] for k in list(r)[:5]})"; }
for v in old new old new; do
  lib=""; [ $v = old ] && lib=$OLD
  MOZJPEG_AMD_LIB=$lib timeout 200 python bench.py --steps 150 --warmup 30 $Q > "$O/metric_$v.log" 2>&1
  echo "metric $v $(tail -1 "$O/metric_$v.log" | line)"
done
for c in c5 c3 c2; do for v in old new; do
  lib=""; [ $v = old ] && lib=$OLD
  MOZJPEG_AMD_LIB=$lib timeout 200 python bench.py --config $c $Q > "$O/${c}_$v.log" 2>&1
  echo "$c $v $(tail -1 "$O/${c}_$v.log" | line)"
done; done
