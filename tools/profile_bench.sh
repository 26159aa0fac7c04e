#!/bin/bash
# rocprofv3 kernel trace of the bench command, serial (MART_OVERLAP_WGRAD=0) and as shipped; summaries -> gpurun_out/
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
for mode in serial overlap; do
  if [ $mode = serial ]; then export MART_OVERLAP_WGRAD=0 MART_TWO_STREAM=0; else export MART_OVERLAP_WGRAD=1 MART_TWO_STREAM=1; fi
  rm -rf gpurun_out/prof_tmp
  timeout 600 rocprofv3 --kernel-trace --stats -d gpurun_out/prof_tmp -- python bench.py --steps 4 --warmup 1 --no-cpu-baseline > gpurun_out/prof_$mode.log 2>&1
  DB=$(find gpurun_out/prof_tmp -name "*.db" | head -1)
  python tools/rocpd_stats.py $DB > gpurun_out/r01_bench_kernel_stats_v17_$mode.csv
  tail -1 gpurun_out/prof_$mode.log | cut -c1-300
  rm -rf gpurun_out/prof_tmp
done
unset MART_OVERLAP_WGRAD MART_TWO_STREAM
timeout 400 python bench.py 2>&1 | tail -1 > gpurun_out/bench_v17.json
