cd $GRAFT_REPO_ROOT
python tests/gpu_stage_check.py 2>&1 | tail -2
python -m pytest tests/test_gpu_parity.py -x -q 2>&1 | tail -3
python tools/prog_scan_times.py 2>&1 | grep -v "^  scan" | tail -6
