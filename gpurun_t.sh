cd $GRAFT_REPO_ROOT
python tests/gpu_stage_check.py 2>&1 | tail -1
python tools/prog_scan_times.py 2>&1 | grep -v "^  scan [1-6][0-9]\|^  scan [4-9] " | tail -14
