#!/bin/bash
# (kept as the record of the A/B: k_front420 and MJH_FRONT_FUSE are NOT in the tree -- profiles/r05q_front_end_fusion_ab.md)
# Round 5: second A/B of the one-kernel front end after both front ends issue all their pixel loads before the first use
cd "$GRAFT_REPO_ROOT" || exit 1
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r5q; mkdir -p "$O"
Q="--no-cpu-baseline --no-host-leg --no-inflight-leg --verify 2 --other-configs none"
for f in 0 1 0 1; do
  MJH_FRONT_FUSE=$f timeout 200 python bench.py --steps 150 --warmup 30 $Q > "$O/metric_$f.log" 2>&1
  echo "fuse=$f $(tail -1 "$O/metric_$f.log" | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'], d['bit_exact']['ok'] if isinstance(d.get('bit_exact'),dict) else d.get('bit_exact'), d['roofline']['kernel_ms_per_call(untimed pass, every kernel bracketed)'])" | cut -c1-330)"
done
for f in 0 1; do
  MJH_FRONT_FUSE=$f timeout 200 python bench.py --config c2 $Q > "$O/c2_$f.log" 2>&1
  echo "c2 fuse=$f $(tail -1 "$O/c2_$f.log" | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'])")"
done
timeout 300 python -m pytest tests/test_gpu_parity.py -q -m gpu -k "one_kernel_front or full_size" -x > "$O/tests.log" 2>&1; tail -2 "$O/tests.log"
