#!/bin/bash
# idle-gap analysis of the shipped (three-stream) schedule: rocprofv3 kernel trace of a short bench run -> tools/rocpd_stats.py
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
rm -rf gpurun_out/prof_tmp
timeout 600 rocprofv3 --kernel-trace -d gpurun_out/prof_tmp -- python bench.py --steps 4 --warmup 1 --no-cpu-baseline --no-kernel-timing > gpurun_out/gaps.log 2>&1
DB=$(find gpurun_out/prof_tmp -name "*.db" | head -1)
python tools/rocpd_stats.py $DB | grep "^#"
rm -rf gpurun_out/prof_tmp
