#!/bin/bash
# usage (on the GPU box, via gpurun): bash tools/gpu_check.sh [tag]  -- the -m gpu suite (new-kernel tests first), smoke(), the default bench line; logs -> gpurun_out/
tag=${1:-check}
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout 600 python -m pytest tests/test_fp16_rows_gpu.py -x -q -m gpu -s > gpurun_out/${tag}_kernels.log 2>&1; echo "kernel tests rc=$?" | tee -a gpurun_out/${tag}_kernels.log
tail -5 gpurun_out/${tag}_kernels.log
timeout 1500 python -m pytest tests -q -m gpu -s --deselect tests/test_fp16_rows_gpu.py > gpurun_out/${tag}_gpu_tests.log 2>&1; echo "gpu suite rc=$?" | tee -a gpurun_out/${tag}_gpu_tests.log
grep -n "passed\|failed\|FAILED\|ERROR" gpurun_out/${tag}_gpu_tests.log | tail -30
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" > gpurun_out/${tag}_smoke.log 2>&1; echo "smoke rc=$?"; tail -2 gpurun_out/${tag}_smoke.log
timeout 600 python bench.py > gpurun_out/${tag}_bench.json 2> gpurun_out/${tag}_bench.err; echo "bench rc=$?"
python - <<PY
import json
try:
    d = json.loads(open("gpurun_out/${tag}_bench.json").read().strip().splitlines()[-1])
    print("BENCH", d["value"], d["unit"], d["ms_per_step"], "ms/step loss", d["loss"], "hits1", d.get("hits1"))
    print("roofline", {k: d["roofline"][k] for k in ("achieved", "frac", "avg_launch_ms", "ms_per_step", "step_frac_of_mfma_peak")})
    for o in d["roofline"]["other_kernels"]:
        print("  ", o["kernel"][:50], o["ms_per_step"], o["achieved"], o["unit"])
    print("parity", json.dumps(d.get("parity"))[:1500])
    print("alt_text_precision", d.get("alt_text_precision"), "alt_head", d.get("alt_entity_head"))
    print("eval", d.get("eval"))
    print("cpu", d.get("cpu_baseline", {}).get("value"))
except Exception as e:
    print("bench parse failed", e)
    print(open("gpurun_out/${tag}_bench.err").read()[-3000:])
PY
