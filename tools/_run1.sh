cd /tmp && export TMPDIR=/tmp; cd "$GRAFT_REPO_ROOT"
timeout 120 python - <<'P' 2>&1 | tail -20
import sys; sys.path.insert(0,'tests'); sys.path.insert(0,'tests/golden')
import torch
from gpu_stage_check import check_case
from cases import images
im = images()
for n in ('testorig','syn96x64'):
    print(n, check_case(im[n], dict(baseline=True, rgb=True), verbose=True))
    print(n, 'notrellis_dc', check_case(im[n], dict(baseline=True, rgb=True, notrellis_dc=True), verbose=True))
P
timeout 200 python -m pytest tests/test_gpu_dropin.py -q -m gpu --timeout 120 -k "rgb_output" 2>&1 | tail -5
