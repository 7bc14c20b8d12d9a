#!/bin/bash
# gpurun_out/<tag>/ (written by tools/profile_round.sh on the GPU box) -> the committed summaries profiles/<tag>_*
# usage: bash tools/collect_profiles.sh r02b [64]
TAG=$1; BATCH=${2:-64}
O=gpurun_out/$TAG; P=profiles/$TAG
db() { find "$O" -name "$1_results.db" | head -1; }
tail -1 "$O/bench_default.log" > "${P}_bench_batch$BATCH.json"
python tools/rocprof_summary.py "$(db stats)" > "${P}_kernel_stats_batch$BATCH.csv"
python tools/pmc_traffic.py "$(db fetch)" "$(db write)" "$BATCH" $((3840*2160*3)) > "${P}_pmc_hbm_traffic_batch$BATCH.json"
python tools/pmc_sq.py "$(db sq)" $(db sq2) --frames=$BATCH > "${P}_pmc_sq_batch$BATCH.json" 2>/dev/null
if [ -f "$O/bench_c3.log" ]; then
  tail -1 "$O/bench_c3.log" > "${P}_c3_bench_batch64.json"
  python tools/rocprof_summary.py "$(db c3_stats)" > "${P}_c3_kernel_stats_batch64.csv"
  python tools/pmc_traffic.py "$(db c3_fetch)" "$(db c3_write)" 64 $((3840*2160*3)) > "${P}_c3_pmc_hbm_traffic_batch64.json"
fi
# traffic of the other configurations: frames per encode call, input bytes per frame (the calibration kernel's input)
for spec in "c2 256 $((1920*1080*3))" "c4 256 $((1920*1080*3))" "c5 1 $((8192*8192*6))" "c5t 1 $((8192*8192*3))"; do
  set -- $spec
  [ -n "$(db $1_fetch)" ] && [ -n "$(db $1_write)" ] && python tools/pmc_traffic.py "$(db $1_fetch)" "$(db $1_write)" $2 $3 > "${P}_$1_pmc_hbm_traffic_batch$2.json"
done
: > "${P}_configs.jsonl"
for c in c2 c4 c5 c5t; do [ -f "$O/bench_$c.log" ] && tail -1 "$O/bench_$c.log" >> "${P}_configs.jsonl"; done
[ -s "${P}_configs.jsonl" ] || rm -f "${P}_configs.jsonl"
[ -n "$(db c5_stats)" ] && python tools/rocprof_summary.py "$(db c5_stats)" > "${P}_c5_kernel_stats.csv"
[ -n "$(db c5t_stats)" ] && python tools/rocprof_summary.py "$(db c5t_stats)" > "${P}_c5t_kernel_stats.csv"
[ -s "$O/dropin.json" ] && cp "$O/dropin.json" "${P}_dropin.json"
ls -la profiles | grep "$TAG"
