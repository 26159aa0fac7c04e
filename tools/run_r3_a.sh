#!/bin/bash
# round-3 GPU run A: error budget on both G7 goldens + full gpu test suite (baseline after the hygiene changes)
mkdir -p gpurun_out
python tools/error_budget.py g7_bench_cond > gpurun_out/budget_cond.txt 2>&1
python tools/error_budget.py g7_bench_plain > gpurun_out/budget_plain.txt 2>&1
timeout 1200 python -m pytest tests -m gpu -q --deselect tests/test_flava_gpu.py::test_flava_vs_reference_at_real_dimensions > gpurun_out/gputests_a.txt 2>&1
tail -5 gpurun_out/gputests_a.txt
cat gpurun_out/budget_cond.txt
