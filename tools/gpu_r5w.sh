#!/bin/bash
# Round 5, last call: the AC trellis walk with BOTH loops as plain divergent loops (the tree) against the library with only the
# scan loop in that form (measured in r5v), alternating; then, with the faster one: parity + fuzz files, the metric's kernel
# trace / traffic / SQ passes (profiles r05w), the default bench line
cd "$GRAFT_REPO_ROOT" || exit 1
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
TAG=r05w; BATCH=64
O=gpurun_out/$TAG; mkdir -p "$O"
Q="--no-cpu-baseline --no-host-leg --no-inflight-leg --verify 2 --other-configs none"
ms() { tail -1 "$1" | python -c "import sys,json; print(json.loads(sys.stdin.read())['ms_per_step'])"; }
SCAN=$PWD/gpurun_ab/libmozjpeg_hip_scan.so
for i in 1 2; do
  MOZJPEG_AMD_LIB=$SCAN timeout 200 python bench.py --steps 150 --warmup 30 $Q > "$O/ab_scan_$i.log" 2>&1
  MOZJPEG_AMD_LIB= timeout 200 python bench.py --steps 150 --warmup 30 $Q > "$O/ab_tree_$i.log" 2>&1
  echo "scan $(ms $O/ab_scan_$i.log)  tree $(ms $O/ab_tree_$i.log)"
done
WIN=$(python -c "
a=($(ms $O/ab_scan_1.log)+$(ms $O/ab_scan_2.log))/2; b=($(ms $O/ab_tree_1.log)+$(ms $O/ab_tree_2.log))/2
print('tree' if b <= a * 1.002 else 'scan')")
echo "winner: $WIN"; echo "$WIN" > "$O/winner.txt"
[ "$WIN" = scan ] && export MOZJPEG_AMD_LIB=$SCAN
echo "== tests"; timeout 400 python -m pytest tests/test_gpu_parity.py tests/test_gpu_fuzz.py -q -m gpu > "$O/tests.log" 2>&1; tail -2 "$O/tests.log"
timeout 200 rocprofv3 --kernel-trace --stats -d "$O" -o stats -- python bench.py --steps 20 --warmup 3 $Q --batch $BATCH > "$O/stats.log" 2>&1
timeout 200 rocprofv3 --pmc FETCH_SIZE --kernel-trace -d "$O" -o fetch -- python bench.py --steps 2 --warmup 1 $Q --batch $BATCH > "$O/fetch.log" 2>&1
timeout 200 rocprofv3 --pmc WRITE_SIZE --kernel-trace -d "$O" -o write -- python bench.py --steps 2 --warmup 1 $Q --batch $BATCH > "$O/write.log" 2>&1
timeout 200 rocprofv3 --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_SALU SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_BUSY_CYCLES --kernel-trace -d "$O" -o sq -- python bench.py --steps 2 --warmup 1 $Q --batch $BATCH > "$O/sq.log" 2>&1
timeout 200 rocprofv3 --pmc SQ_THREAD_CYCLES_VALU SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_ANY SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR GRBM_GUI_ACTIVE --kernel-trace -d "$O" -o sq2 -- python bench.py --steps 2 --warmup 1 $Q --batch $BATCH > "$O/sq2.log" 2>&1
if [ "$WIN" = tree ]; then : > "$O/bench_default.log"; bash tools/collect_profiles.sh $TAG $BATCH > "$O/collect.log" 2>&1; fi
echo "== bench"; timeout 400 python bench.py > "$O/bench_default.log" 2>&1; tail -1 "$O/bench_default.log" | cut -c1-330
