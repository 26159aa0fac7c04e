#!/bin/bash
# rocprofv3 kernel trace of the evaluation passes (bf16, then fp32-accurate); summary -> gpurun_out/$1 (default eval_fp32_kernel_stats.csv)
out=${1:-eval_fp32_kernel_stats.csv}
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
rm -rf gpurun_out/prof_tmp
timeout 600 rocprofv3 --kernel-trace --stats -d gpurun_out/prof_tmp -- python tools/eval_fp32_profile.py 2 > gpurun_out/eval_fp32_profile.log 2>&1
tail -3 gpurun_out/eval_fp32_profile.log
DB=$(find gpurun_out/prof_tmp -name "*.db" | head -1)
python tools/rocpd_stats.py $DB > gpurun_out/$out
head -24 gpurun_out/$out | cut -c1-170
rm -rf gpurun_out/prof_tmp
