#!/bin/bash
cd $GRAFT_REPO_ROOT
for rep in 1 2; do
  echo "== shipped"; timeout 300 python tools/bench_attn.py 2>&1 | grep "attn_bwd"
  for l in "$@"; do echo "== $l"; MART_HIP_LIB=$PWD/$l timeout 300 python tools/bench_attn.py 2>&1 | grep "attn_bwd"; done
done
