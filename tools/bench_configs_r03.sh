#!/bin/bash
# The other BASELINE configurations (parity cases, not the headline): bench lines + rocprofv3 kernel-trace summaries, kept under profiles/.
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
run() { # tag, bench args...
  tag=$1; shift
  timeout 600 python bench.py "$@" --no-cpu-baseline 2>/dev/null | tail -1 > gpurun_out/r03_bench_$tag.json
  rm -rf gpurun_out/prof_tmp
  MART_OVERLAP_WGRAD=0 MART_TWO_STREAM=0 timeout 600 rocprofv3 --kernel-trace --stats -d gpurun_out/prof_tmp -- python bench.py "$@" --steps 3 --warmup 1 --no-cpu-baseline --no-kernel-timing > /dev/null 2>&1
  DB=$(find gpurun_out/prof_tmp -name "*.db" | head -1)
  python tools/rocpd_stats.py $DB > gpurun_out/r03_kernel_stats_${tag}_serial.csv
  rm -rf gpurun_out/prof_tmp
  python -c "import json; d=json.load(open('gpurun_out/r03_bench_$tag.json')); print('$tag', d['value'], d['unit'], d['ms_per_step'], 'ms/step', d['roofline']['step_frac_of_mfma_peak'])"
}
run p49 --patch 32
run head2063 --entity-head 2063
run pretrain --task pretrain --seq-len 96
run pretrain_p49 --task pretrain --seq-len 96 --patch 32
run flava --model flava --batch 128
