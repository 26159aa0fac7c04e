#!/bin/bash
# A/B of the backward pass's queue switches; logs -> gpurun_out/
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
for args in "--patch 32" "--patch 16"; do for r in 1 2; do for v in 0 1; do
  echo -n "[$args] WGRAD_ALT=$v: "
  MART_WGRAD_ALT=$v timeout 300 python bench.py $args --steps 20 --warmup 4 --no-cpu-baseline --no-kernel-timing --train-only 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['ms_per_step'], d['value'], d.get('loss'))"
done; done; done 2>&1 | tee gpurun_out/ab_wgrad_alt.txt
