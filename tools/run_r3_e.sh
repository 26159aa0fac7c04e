#!/bin/bash
mkdir -p gpurun_out
timeout 1500 python -m pytest tests -m gpu -q > gpurun_out/gputests_e.txt 2>&1
tail -6 gpurun_out/gputests_e.txt
bash tools/profile_r03.sh r03a 2>&1 | tail -5
