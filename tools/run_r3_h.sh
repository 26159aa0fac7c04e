#!/bin/bash
# round 3 (h): fused attention backward with the Q / dO prefetch as opaque assembly (no compiler-forced vmcnt(0) behind the issue)
cd $GRAFT_REPO_ROOT
O=gpurun_out/r3h; mkdir -p $O
timeout 600 python -m pytest tests/test_kernels_gpu.py -q -x -k "attention" 2>&1 | tail -3 > $O/tests.log
echo "--- two-kernel backward, shipped vs STAGE_RAW" > $O/ab.log
for rep in 1 2; do
  for l in "" tools/variants/attn_stageraw.so; do
    echo "== ${l:-shipped}" >> $O/ab.log
    MART_ATTN_FUSED=0 MART_HIP_LIB=${l:+$PWD/$l} timeout 300 python tools/bench_attn.py 2>&1 | grep "attn_bwd" >> $O/ab.log
  done
done
bash tools/ab_step.sh "MART_HIP_LIB=$PWD/tools/variants/attn_noraw.so" "MART_X=1" > $O/step.log 2>&1
