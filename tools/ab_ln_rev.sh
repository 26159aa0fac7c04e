#!/bin/bash
# MART_LN_REV A/B (tools/mall_probe.py, then the training step in alternation); logs -> gpurun_out/
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
timeout 300 python tools/mall_probe.py 20 > gpurun_out/mall_probe.txt 2>&1; tail -14 gpurun_out/mall_probe.txt
for r in 1 2; do for v in 0 3 1 2; do
  echo -n "MART_LN_REV=$v: "
  MART_LN_REV=$v timeout 300 python bench.py --steps 12 --warmup 3 --no-cpu-baseline --no-kernel-timing --train-only 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['ms_per_step'], d['value'], d.get('loss'))"
done; done 2>&1 | tee gpurun_out/ab_ln_rev.txt
