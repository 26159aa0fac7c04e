#!/bin/bash
cd $GRAFT_REPO_ROOT
run() { timeout 300 python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-kernel-timing 2>&1 | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('   ', d['value'], d['ms_per_step'], d['loss'])"; }
echo "== single stream"; run; run
echo "== two streams"; export MART_TWO_STREAM=1; run; run; run
echo "== tests with two streams"; timeout 1500 python -m pytest tests/test_model_gpu.py tests/test_mars_dev_gpu.py -x -q 2>&1 | tail -2
