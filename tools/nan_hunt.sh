#!/bin/bash
cd $GRAFT_REPO_ROOT

for i in 1 2 3 4 5 6; do echo "== run $i"; timeout 400 python tools/debug_nan.py 256 16 150 10 2>&1 | grep -E "^step" | tail -1 | cut -c1-200; done
timeout 2000 python -m pytest tests -m gpu -x -q 2>&1 | tail -2
