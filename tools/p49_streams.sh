#!/bin/bash
# rocprofv3 kernel trace of the shipped multi-queue schedule -> per-queue busy time and the main queue's gaps (tools/rocpd_streams.py); usage: bash tools/p49_streams.sh <out name> [bench args]
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out; rm -rf gpurun_out/prof_tmp
OUT=$1; shift
timeout 600 rocprofv3 --kernel-trace --stats -d gpurun_out/prof_tmp -- python bench.py "$@" --steps 4 --warmup 10 --no-cpu-baseline --train-only --no-kernel-timing > gpurun_out/prof_streams.log 2>&1
DB=$(find gpurun_out/prof_tmp -name "*.db" | head -1)
python tools/rocpd_streams.py $DB > gpurun_out/$OUT 2>&1
rm -rf gpurun_out/prof_tmp; head -40 gpurun_out/$OUT | cut -c1-200
