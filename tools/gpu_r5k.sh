#!/bin/bash
# Round 5: the whole GPU suite on the tree with scan scripts / sampling factors / the new fuzz family, then the parity +
# host-path + drop-in files under the memory fence (MJH_GUARD=2)
cd "$GRAFT_REPO_ROOT" || exit 1
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r5k; mkdir -p "$O"
echo "== 1. the whole suite"; timeout 800 python -m pytest tests -q -m gpu > "$O/suite.log" 2>&1; tail -6 "$O/suite.log"
echo "== 2. MJH_GUARD=2"; MJH_GUARD=2 timeout 600 python -m pytest tests/test_gpu_parity.py tests/test_gpu_host_path.py tests/test_gpu_fuzz.py -q -m gpu -x > "$O/guard2.log" 2>&1; tail -3 "$O/guard2.log"
