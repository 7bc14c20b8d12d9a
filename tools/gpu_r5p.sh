#!/bin/bash
# (kept as the record of the A/B: k_front420 and MJH_FRONT_FUSE are NOT in the tree -- profiles/r05q_front_end_fusion_ab.md)
# Round 5: the one-kernel front end (k_front420: pixel rows -> coefficients) on the chip: its tests, A/B against the two-kernel
# front end on the metric / C2 / C3, kernel stats of both
cd "$GRAFT_REPO_ROOT" || exit 1
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r5p; mkdir -p "$O"
timeout 400 python -m pytest tests/test_gpu_parity.py -q -m gpu -k "one_kernel_front or full_size or every_stage" -x > "$O/tests.log" 2>&1; tail -3 "$O/tests.log"
Q="--no-cpu-baseline --no-host-leg --no-inflight-leg --verify 2 --other-configs none"
for f in 0 1 0 1; do
  MJH_FRONT_FUSE=$f timeout 200 python bench.py --steps 150 --warmup 30 $Q > "$O/metric_$f.log" 2>&1
  echo "fuse=$f $(tail -1 "$O/metric_$f.log" | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'], d['bit_exact']['ok'] if isinstance(d.get('bit_exact'),dict) else d.get('bit_exact'), d['roofline']['kernel_ms_per_call(untimed pass, every kernel bracketed)'])")"
done
for c in c2 c3; do for f in 0 1; do
  MJH_FRONT_FUSE=$f timeout 200 python bench.py --config $c $Q > "$O/${c}_$f.log" 2>&1
  echo "$c fuse=$f $(tail -1 "$O/${c}_$f.log" | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'])")"
done; done
timeout 200 rocprofv3 --kernel-trace --stats -d "$O" -o fused_stats -- python bench.py --steps 5 --warmup 2 $Q > "$O/fused_stats.log" 2>&1
f=$(find "$O" -name "fused_stats_results.db" | head -1); [ -n "$f" ] && python tools/rocprof_summary.py "$f" | head -8
