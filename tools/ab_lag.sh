#!/bin/bash
# A/B of queue switches on top of engine.wgrad_lag / wgrad_two: MART_STREAM_OPT, MART_OPT_PRIO; logs -> gpurun_out/
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
for args in "--patch 16" "--patch 32"; do for r in 1 2; do for cfg in "1 0" "0 0" "1 -1"; do
  set -- $cfg
  echo -n "[$args] STREAM_OPT=$1 OPT_PRIO=$2: "
  MART_STREAM_OPT=$1 MART_OPT_PRIO=$2 timeout 300 python bench.py $args --steps 20 --warmup 4 --no-cpu-baseline --no-kernel-timing --train-only 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['ms_per_step'], d['value'], d.get('loss'))"
done; done; done 2>&1 | tee gpurun_out/ab_stream_opt.txt
