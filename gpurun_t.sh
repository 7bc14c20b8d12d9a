cd $GRAFT_REPO_ROOT
python tests/gpu_stage_check.py 2>&1 | tail -1
python -m pytest tests -m gpu -x -q 2>&1 | tail -2
python bench.py --steps 20 --warmup 3 --no-cpu-baseline --batch 16 2>&1 | tail -1 > gpurun_out/bench_last.json
python -c "
import json; d=json.load(open('gpurun_out/bench_last.json')); print(d['value'], d['ms_per_step'], d['bit_exact_vs_oracle'], d['roofline']['kernel'], d['roofline']['kernel_ms'], d['roofline']['frac'])
print(d['roofline']['kernel_ms_per_step(untimed pass, every kernel bracketed)'])"
