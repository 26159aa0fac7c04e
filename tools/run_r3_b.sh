#!/bin/bash
# round-3 GPU run B: parity tests (head split default, text split mode, G8b, G9)
mkdir -p gpurun_out
timeout 1500 python -m pytest tests/test_parity_full_gpu.py tests/test_flava_gpu.py tests/test_model_gpu.py -m gpu -q -s > gpurun_out/parity_r3b.txt 2>&1
tail -15 gpurun_out/parity_r3b.txt
grep -n "max|dlogit|\|control\|text_split\|loss hip" gpurun_out/parity_r3b.txt | head -60
