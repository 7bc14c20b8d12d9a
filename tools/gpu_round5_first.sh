#!/bin/bash
# Round 5's first GPU call (round 4 ended without GPU minutes for the tile-sorted planes of mjh_sorted.hip):
#   1. the parity tests of the default path (its kernels are the machine code that passed the suite in round 4);
#   2. the tile-sorted planes on the chip for the first time: their own test, then the whole parity file with every batch sorted;
#   3. A/B of MJH_SORTED_UQ on the metric workload (files must be identical) and the guard's unmapped-page mode on the sorted path;
#   4. the contract bench line.
# usage: gpurun --timeout 1500 -- 'bash tools/gpu_round5_first.sh'      (about 20 GPU-minutes)
cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/r5first; mkdir -p "$O"
echo "== 1. default path"; timeout 600 python -m pytest tests/test_gpu_parity.py -x -q -m gpu > "$O/parity_default.log" 2>&1; tail -2 "$O/parity_default.log"
echo "== 2a. sorted planes: own test"; MJH_TEST_SORTED=1 timeout 300 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k tile_sorted > "$O/sorted_own.log" 2>&1; tail -2 "$O/sorted_own.log"
echo "== 2b. every batch sorted"; MJH_TEST_SORTED=1 MJH_SORTED_UQ=2 timeout 600 python -m pytest tests/test_gpu_parity.py -x -q -m gpu > "$O/parity_sorted.log" 2>&1; tail -2 "$O/parity_sorted.log"
echo "== 2b'. every batch sorted, tiles of 512 and 128"; for t in 512 128; do MJH_TEST_SORTED=1 MJH_SORTED_UQ=2 MJH_SORTED_TILE=$t timeout 600 python -m pytest tests/test_gpu_parity.py -x -q -m gpu > "$O/parity_sorted_$t.log" 2>&1; tail -1 "$O/parity_sorted_$t.log"; done
echo "== 2c. sorted under the memory fence"; MJH_TEST_SORTED=1 MJH_SORTED_UQ=2 MJH_GUARD=2 timeout 300 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "tile_sorted or every_stage" > "$O/sorted_guard.log" 2>&1; tail -2 "$O/sorted_guard.log"
echo "== 2d. queue records from the FDCT kernel: own test, then the parity file with the mode on"; MJH_TEST_SORTED=1 timeout 300 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k queue_records > "$O/rec_own.log" 2>&1; tail -1 "$O/rec_own.log"
MJH_TEST_SORTED=1 MJH_TRELLIS_REC=1 timeout 600 python -m pytest tests/test_gpu_parity.py -x -q -m gpu > "$O/parity_rec.log" 2>&1; tail -1 "$O/parity_rec.log"
echo "== 3a. A/B of the record mode on the metric workload"; timeout 400 python tools/bench_variants.py --env MJH_TRELLIS_REC --variants 0,1 --steps 10 > "$O/variants_rec.log" 2>&1; grep '^{' "$O/variants_rec.log" | cut -c1-520
echo "== 3a'. the record mode on C3 and C5t"; for c in c3 c5t; do timeout 400 python tools/bench_variants.py --config $c --env MJH_TRELLIS_REC --variants 0,1 --steps 6 > "$O/variants_rec_$c.log" 2>&1; grep '^{' "$O/variants_rec_$c.log" | cut -c1-400; done
echo "== 3a''. the record mode with tiles of 512 blocks (eight passes: the gathers are rows of records now, not 63 planes)"; MJH_TRELLIS_REC=1 timeout 400 python tools/bench_variants.py --env MJH_TRELLIS_V3 --variants 4,8 --steps 10 > "$O/variants_rec_v3.log" 2>&1; grep '^{' "$O/variants_rec_v3.log" | cut -c1-400
echo "== 3. A/B on the metric workload"; timeout 400 python tools/bench_variants.py --env MJH_SORTED_UQ --variants 0,1 --steps 10 > "$O/variants.log" 2>&1; grep '^{' "$O/variants.log" | cut -c1-520
echo "== 3b. tile sizes (sorted on)"; MJH_SORTED_UQ=1 timeout 400 python tools/bench_variants.py --env MJH_SORTED_TILE --variants 128,256,512 --steps 10 > "$O/variants_tile.log" 2>&1; grep '^{' "$O/variants_tile.log" | cut -c1-520
echo "== 3c. MJH_PP_SKIPLOW (mjh_prog_sl.hip): own test, then A/B on C3"; MJH_TEST_SORTED=1 timeout 300 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k skiplow > "$O/skiplow_own.log" 2>&1; tail -1 "$O/skiplow_own.log"
timeout 400 python tools/bench_variants.py --config c3 --env MJH_PP_SKIPLOW --variants 0,1 --steps 6 > "$O/variants_skiplow_c3.log" 2>&1; grep '^{' "$O/variants_skiplow_c3.log" | cut -c1-400
echo "== 4. bench"; timeout 300 python bench.py --no-cpu-baseline --no-host-leg --no-inflight-leg > "$O/bench.log" 2>&1; tail -1 "$O/bench.log" | cut -c1-600
