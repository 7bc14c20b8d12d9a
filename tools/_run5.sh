cd /tmp && export TMPDIR=/tmp; cd "$GRAFT_REPO_ROOT"
timeout 900 python -m pytest tests/test_gpu_parity.py -q -m gpu --timeout 120 -k "q_opt or eob_opt or scans_in_trellis or all_trellis or dc_ver or dc_scan" > gpurun_out/t5.log 2>&1; tail -30 gpurun_out/t5.log
