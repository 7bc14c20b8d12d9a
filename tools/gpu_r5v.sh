#!/bin/bash
# Round 5: the trellis walks' scan loop as a plain divergent loop (no ballot-driven loop with a bypass block) against the
# library of commit 2b59fb1, alternating on one box; then the trellis-heavy GPU tests on the new library
cd "$GRAFT_REPO_ROOT" || exit 1
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r5v; mkdir -p "$O"
Q="--no-cpu-baseline --no-host-leg --no-inflight-leg --verify 2 --other-configs none"
line() { python -c "import sys,json; d=json.loads(sys.stdin.read()); r=d['roofline']['kernel_ms_per_call(untimed pass, every kernel bracketed)']; print(d['value'], d['ms_per_step'], d['bit_exact']['ok'] if isinstance(d.get('bit_exact'),dict) else d.get('bit_exact'), {k: r[k] for k in list(r)[:4]})"; }
for v in 2b59fb1 new 2b59fb1 new; do
  lib=""; [ $v != new ] && lib=$PWD/gpurun_ab/libmozjpeg_hip_$v.so
  MOZJPEG_AMD_LIB=$lib timeout 200 python bench.py --steps 150 --warmup 30 $Q > "$O/metric_$v.log" 2>&1
  echo "metric $v $(tail -1 "$O/metric_$v.log" | line)"
done
for c in c5t c3 c2; do for v in 2b59fb1 new; do
  lib=""; [ $v != new ] && lib=$PWD/gpurun_ab/libmozjpeg_hip_$v.so
  MOZJPEG_AMD_LIB=$lib timeout 200 python bench.py --config $c $Q > "$O/${c}_$v.log" 2>&1
  echo "$c $v $(tail -1 "$O/${c}_$v.log" | line)"
done; done
timeout 300 python -m pytest tests/test_gpu_parity.py -q -m gpu -x -k "every_stage or first_tier or full_size or q_opt or eob" > "$O/tests.log" 2>&1; tail -2 "$O/tests.log"
