#!/bin/bash
# usage: tools/ab_attn.sh lib1.so lib2.so ...   -- tools/bench_attn.py on the shipped library and on each variant (same box, interleaved twice)
cd $GRAFT_REPO_ROOT
for rep in 1 2; do
  echo "== shipped"; timeout 300 python tools/bench_attn.py 2>&1 | grep "attn_"
  for l in "$@"; do echo "== $l"; MART_HIP_LIB=$PWD/$l timeout 300 python tools/bench_attn.py 2>&1 | grep "attn_"; done
done
