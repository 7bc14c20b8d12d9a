#!/bin/bash
# How profiles/rNN_* are produced (run on the GPU box: /usr/local/graft/bin/gpurun -- 'bash tools/profile_round.sh r01x'):
# the bench line, the rocprofv3 kernel-trace summary of the same command, and the PMC passes -- each counter set in
# its own run with --kernel-trace only, as /opt/skills/guides/MI355X_MICROARCH.md prescribes.  Results land in
# gpurun_out/<tag>/ and are turned into profiles/<tag>_* by tools/rocprof_summary.py and tools/pmc_traffic.py.
TAG=${1:-r01x}
BATCH=${2:-64}
cd /tmp && export TMPDIR=/tmp
cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/$TAG; mkdir -p "$O"
python bench.py > "$O/bench_default.log" 2>&1; tail -1 "$O/bench_default.log" | cut -c1-300
rocprofv3 --kernel-trace --stats -d "$O" -o stats -- python bench.py --steps 20 --warmup 3 --no-cpu-baseline --batch "$BATCH" > "$O/stats.log" 2>&1
rocprofv3 --pmc FETCH_SIZE --kernel-trace -d "$O" -o fetch -- python bench.py --steps 2 --warmup 1 --no-cpu-baseline --batch "$BATCH" > "$O/fetch.log" 2>&1
rocprofv3 --pmc WRITE_SIZE --kernel-trace -d "$O" -o write -- python bench.py --steps 2 --warmup 1 --no-cpu-baseline --batch "$BATCH" > "$O/write.log" 2>&1
rocprofv3 --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_SALU SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_BUSY_CYCLES --kernel-trace -d "$O" -o sq -- python bench.py --steps 2 --warmup 1 --no-cpu-baseline --batch "$BATCH" > "$O/sq.log" 2>&1
if [ "$3" = "all" ]; then
  python tools/bench_configs.py > "$O/configs.jsonl" 2> "$O/configs.err"
  python tools/bench_host.py --batch 16 > "$O/host.json" 2> "$O/host.err"; cat "$O/host.json"
fi
