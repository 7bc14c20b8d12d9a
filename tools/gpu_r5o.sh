#!/bin/bash
# Round 5: trellis_q_opt with the arithmetic coder for any number of loops -- the new goldens and the seeded family on the chip (also
# under MJH_GUARD=2), the arithmetic tests around them, a 1080p frame against the reference binary
cd "$GRAFT_REPO_ROOT" || exit 1
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r5o; mkdir -p "$O"
timeout 600 python -m pytest tests/test_gpu_parity.py tests/test_gpu_fuzz.py -q -m gpu -k "q_opt or arith" > "$O/arith_qopt.log" 2>&1; tail -3 "$O/arith_qopt.log"
MJH_GUARD=2 timeout 600 python -m pytest tests/test_gpu_parity.py tests/test_gpu_fuzz.py -q -m gpu -k "arith and q_opt" > "$O/arith_qopt_guard.log" 2>&1; tail -3 "$O/arith_qopt_guard.log"
timeout 300 python - > "$O/big.log" 2>&1 <<'PY'
import sys, numpy as np
sys.path.insert(0, "tests")
import oracle_lib as O, mozjpeg_amd as M
for kw in (dict(arithmetic=True, baseline=True, trellis_q_opt=True, trellis_loops=2), dict(arithmetic=True, baseline=True, gray=True, trellis_q_opt=True),
           dict(arithmetic=True, trellis_q_opt=True, trellis_loops=4, quality=85)):
    w, h = 1920, 1080
    img = O.synthetic_frame(w, h, 77)
    ref = O.ref_encode(img, **kw)[0]
    enc = M.Encoder(M.make_params(w, h, **kw), max_batch=2)
    got = enc.encode_host(np.stack([img, img]))
    enc.close()
    print(kw, len(ref), len(got[0]), "IDENTICAL" if got[0] == ref and got[1] == ref else "DIFFERENT")
PY
cat "$O/big.log"
