#!/bin/bash
mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_kernels_gpu.py -m gpu -q -k "attention or attn" 2>&1 | tail -3
for i in 1 2; do python tools/bench_attn.py 2>&1 | grep attn_; done
