#!/bin/bash
# Round 5, final tree: smoke(), the parity / fuzz files under MJH_GUARD=2 (an unmapped page behind every device buffer), drop-in throughput
cd "$GRAFT_REPO_ROOT" || exit 1
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r05u; mkdir -p "$O"
timeout 200 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" > "$O/smoke.log" 2>&1; tail -1 "$O/smoke.log"
MJH_GUARD=2 timeout 600 python -m pytest tests/test_gpu_parity.py tests/test_gpu_fuzz.py -q -m gpu -x > "$O/guard2.log" 2>&1; tail -2 "$O/guard2.log"
timeout 400 python tools/bench_dropin.py > "$O/dropin.json" 2> "$O/dropin.err"; tail -4 "$O/dropin.err"; python -c "
import json; d=json.load(open('$O/dropin.json'))
for r in d['library_client']: print(r['mode'], r['threads'], r['images_per_s'], r['mpix_per_s'])
"
