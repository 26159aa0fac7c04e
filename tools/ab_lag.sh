#!/bin/bash
# A/B of the scheduling switches of the backward pass: MART_WGRAD_LAG (no per-layer join of the weight-gradient stream) x MART_TEXT_PRIO; logs -> gpurun_out/
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
for args in "--patch 16" "--patch 32" "--task pretrain --seq-len 96"; do for r in 1 2 3; do for cfg in "0 0" "1 0" "1 -1"; do
  set -- $cfg
  echo -n "[$args] LAG=$1 TEXT_PRIO=$2: "
  MART_WGRAD_LAG=$1 MART_TEXT_PRIO=$2 timeout 300 python bench.py $args --steps 12 --warmup 3 --no-cpu-baseline --no-kernel-timing --train-only 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['ms_per_step'], d['value'], d.get('loss'))"
done; done; done 2>&1 | tee gpurun_out/ab_wgrad_lag2.txt
