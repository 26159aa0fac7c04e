#!/bin/bash
cd $GRAFT_REPO_ROOT
for rep in 1 2; do
  echo "== shipped TPW2"; timeout 300 python tools/bench_attn.py 2>&1 | grep "attn_fwd"
  echo "== shipped TPW1 (168 vgpr)"; MART_ATTN_TPW=1 timeout 300 python tools/bench_attn.py 2>&1 | grep "attn_fwd"
  for w in 3 4; do echo "== TPW1 minw$w"; MART_ATTN_TPW=1 MART_HIP_LIB=$PWD/tools/variants/attn_minw$w.so timeout 300 python tools/bench_attn.py 2>&1 | grep "attn_fwd"; done
done
