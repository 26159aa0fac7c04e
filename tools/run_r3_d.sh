#!/bin/bash
mkdir -p gpurun_out
timeout 1500 python -m pytest tests/test_parity_full_gpu.py -m gpu -q -s -k "fp32_training_step" > gpurun_out/fp32_train.txt 2>&1
grep -n "fp32 training step\|Error\|error\|assert\|passed\|failed" gpurun_out/fp32_train.txt | head -40
tail -30 gpurun_out/fp32_train.txt
