#!/bin/bash
# round-3 evidence run: smoke, gpu tests, bench + rocprof + PMC (tools/profile_r03.sh), the other configurations on the same box
mkdir -p gpurun_out
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -2
timeout 1500 python -m pytest tests -m gpu -q > gpurun_out/gputests_final.txt 2>&1; tail -3 gpurun_out/gputests_final.txt
bash tools/profile_r03.sh r03 2>&1 | tail -3
bash tools/bench_configs_r03.sh 2>&1 | tail -6
