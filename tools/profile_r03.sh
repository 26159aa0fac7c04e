#!/bin/bash
# Round-3 evidence, one gpurun call: the bench line, rocprofv3 kernel-trace summaries (serial + shipped schedule), PMC passes of the
# dominant kernel.  Everything lands in gpurun_out/; the summaries that are judged are copied to profiles/ and committed.
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
TAG=${1:-r03}
timeout 500 python bench.py 2> gpurun_out/${TAG}_bench.err | tail -1 > gpurun_out/${TAG}_bench.json
for mode in serial overlap; do
  if [ $mode = serial ]; then export MART_OVERLAP_WGRAD=0 MART_TWO_STREAM=0; else export MART_OVERLAP_WGRAD=1 MART_TWO_STREAM=1; fi
  rm -rf gpurun_out/prof_tmp
  timeout 600 rocprofv3 --kernel-trace --stats -d gpurun_out/prof_tmp -- python bench.py --steps 4 --warmup 1 --no-cpu-baseline --train-only > gpurun_out/prof_$mode.log 2>&1
  DB=$(find gpurun_out/prof_tmp -name "*.db" | head -1)
  python tools/rocpd_stats.py $DB > gpurun_out/${TAG}_bench_kernel_stats_$mode.csv
  tail -1 gpurun_out/prof_$mode.log | cut -c1-200
  rm -rf gpurun_out/prof_tmp
done
export MART_OVERLAP_WGRAD=0 MART_TWO_STREAM=0
CMD="python bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-kernel-timing"
bash tools/pmc.sh "gemm_nt_kernel<256" gpurun_out/${TAG}_pmc_gemm_nt.txt -- $CMD > /dev/null 2>&1
python tools/pmc_to_json.py gpurun_out/${TAG}_pmc_gemm_nt.txt gpurun_out/${TAG}_pmc_gemm_nt.json "MART_OVERLAP_WGRAD=0 MART_TWO_STREAM=0 rocprofv3 --pmc <group> -- $CMD (one pass per counter group: tools/pmc.sh)" | cut -c1-300
bash tools/pmc.sh "gemm_tn8" gpurun_out/${TAG}_pmc_gemm_tn.txt -- $CMD > /dev/null 2>&1
unset MART_OVERLAP_WGRAD MART_TWO_STREAM
cat gpurun_out/${TAG}_bench.json | cut -c1-1500
# attention PMC (vision forward at four waves per SIMD, fused backward)
bash tools/pmc.sh "attn_fwd_k" gpurun_out/${TAG}_pmc_attn_fwd.txt -- python tools/bench_attn.py > /dev/null 2>&1
bash tools/pmc.sh "attn_bwd_fused" gpurun_out/${TAG}_pmc_attn_bwd_fused.txt -- python tools/bench_attn.py > /dev/null 2>&1
