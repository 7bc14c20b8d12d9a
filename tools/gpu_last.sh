#!/bin/bash
# last call of the round: the arithmetic tests and bench lines on the final tree, then the whole GPU suite
cd "$GRAFT_REPO_ROOT" || exit 1
ARITH_STEPS=1 bash tools/gpu_arith.sh 2>&1 | tail -7
timeout 200 python -m pytest tests -q -m gpu -n 6 -p no:cacheprovider 2>&1 | tail -3
