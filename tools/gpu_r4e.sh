#!/bin/bash
# round 4, last tree: the whole GPU suite first (stops here when anything fails), then the arithmetic bench lines, then the
# profile round of every configuration
cd "$GRAFT_REPO_ROOT" || exit 1
TAG=${1:-r04e}
set -o pipefail
timeout 900 python -m pytest tests -q -m gpu -n 4 -p no:cacheprovider 2>&1 | tail -3 || { echo "SUITE FAILED"; exit 1; }
ARITH_SKIP_TESTS=1 ARITH_STEPS=2 bash tools/gpu_arith.sh 2>&1 | tail -6
bash tools/profile_round.sh "$TAG" 64 all
