cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out; rm -rf gpurun_out/prof_tmp
timeout 600 rocprofv3 --kernel-trace --stats -d gpurun_out/prof_tmp -- python bench.py --patch 32 --steps 4 --warmup 1 --no-cpu-baseline --train-only --no-kernel-timing > gpurun_out/prof_p49.log 2>&1
DB=$(find gpurun_out/prof_tmp -name "*.db" | head -1)
python tools/rocpd_streams.py $DB > gpurun_out/r05_streams_overlap_p49.txt 2>&1
rm -rf gpurun_out/prof_tmp; head -8 gpurun_out/r05_streams_overlap_p49.txt
