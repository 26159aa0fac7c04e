#!/bin/bash
# usage (gpurun): bash tools/tail_trace.sh <out name> [bench args]   -- kernel-by-kernel listing of the step's tail (tools/rocpd_tail.py)
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out; rm -rf gpurun_out/prof_tmp
OUT=$1; shift
timeout 600 rocprofv3 --kernel-trace -d gpurun_out/prof_tmp -- python bench.py "$@" --steps 4 --warmup 2 --no-cpu-baseline --train-only --no-kernel-timing > gpurun_out/prof_tail.log 2>&1
DB=$(find gpurun_out/prof_tmp -name "*.db" | head -1)
python tools/rocpd_tail.py $DB 700 1500 > gpurun_out/$OUT 2>&1
rm -rf gpurun_out/prof_tmp; wc -l gpurun_out/$OUT
