cd $GRAFT_REPO_ROOT
python tests/gpu_stage_check.py 2>&1 | tail -1
python -m pytest tests/test_gpu_parity.py -x -q 2>&1 | tail -3
python tools/prog_scan_times.py 2>&1 | grep -v "^  scan [2-6][0-9]\|^  scan 1[3-9]" | tail -40
