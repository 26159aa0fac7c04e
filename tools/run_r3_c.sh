#!/bin/bash
# round-3 GPU run C: bench (default), two-rank flow, rocprof kernel stats
mkdir -p gpurun_out
python bench.py --steps 20 --warmup 5 > gpurun_out/bench_r3c.json 2> gpurun_out/bench_r3c.err
tail -3 gpurun_out/bench_r3c.err
cat gpurun_out/bench_r3c.json
timeout 900 python -m pytest tests/test_two_ranks_one_gpu.py tests/test_ddp_equivalence_gpu.py tests/test_ddp_nccl_gpu.py -m gpu -q -x 2>&1 | tail -5
