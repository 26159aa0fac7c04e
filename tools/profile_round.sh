#!/bin/bash
# Evidence of a round's final state, one gpurun call: the bench line, rocprofv3 kernel-trace summaries (serial + shipped schedule), PMC passes of the
# dominant kernel (-> roofline.traffic), the other BASELINE configurations.  Everything lands in gpurun_out/; what is judged is copied to profiles/.
# usage: bash tools/profile_round.sh r06
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
TAG=${1:-r06}
mkdir -p gpurun_out
for mode in serial overlap; do
  if [ $mode = serial ]; then export MART_OVERLAP_WGRAD=0 MART_TWO_STREAM=0; else export MART_OVERLAP_WGRAD=1 MART_TWO_STREAM=1; fi
  rm -rf gpurun_out/prof_tmp
  timeout 600 rocprofv3 --kernel-trace --stats -d gpurun_out/prof_tmp -- python bench.py --steps 4 --warmup 1 --no-cpu-baseline --train-only > gpurun_out/prof_$mode.log 2>&1
  DB=$(find gpurun_out/prof_tmp -name "*.db" | head -1)
  python tools/rocpd_stats.py $DB > gpurun_out/${TAG}_bench_kernel_stats_$mode.csv
  [ $mode = overlap ] && python tools/rocpd_streams.py $DB > gpurun_out/${TAG}_streams_overlap.txt 2>&1
  tail -1 gpurun_out/prof_$mode.log | cut -c1-200
  rm -rf gpurun_out/prof_tmp
done
export MART_OVERLAP_WGRAD=0 MART_TWO_STREAM=0
CMD="python bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-kernel-timing --train-only"
bash tools/pmc.sh "gemm_nt_kernel<256" gpurun_out/${TAG}_pmc_gemm_nt.txt -- $CMD > /dev/null 2>&1
python tools/pmc_to_json.py gpurun_out/${TAG}_pmc_gemm_nt.txt gpurun_out/${TAG}_pmc_gemm_nt.json "MART_OVERLAP_WGRAD=0 MART_TWO_STREAM=0 rocprofv3 --pmc <group> -- $CMD (one pass per counter group: tools/pmc.sh)" | cut -c1-300
bash tools/pmc.sh "gemm_tn8" gpurun_out/${TAG}_pmc_gemm_tn.txt -- $CMD > /dev/null 2>&1
bash tools/pmc.sh "attn_fwd_k<false" gpurun_out/${TAG}_pmc_attn_fwd.txt -- $CMD > /dev/null 2>&1
bash tools/pmc.sh "attn_bwd_fused" gpurun_out/${TAG}_pmc_attn_bwd_fused.txt -- $CMD > /dev/null 2>&1
unset MART_OVERLAP_WGRAD MART_TWO_STREAM
# the bench line last of the three, so that its roofline.traffic is THIS run's PMC file (bench.py reads profiles/<TAG>_pmc_gemm_nt.json and checks the source hash)
cp gpurun_out/${TAG}_pmc_gemm_nt.json profiles/${TAG}_pmc_gemm_nt.json
timeout 600 python bench.py 2> gpurun_out/${TAG}_bench.err | tail -1 > gpurun_out/${TAG}_bench.json
run() { # tag, bench args...
  tag=$1; shift
  timeout 600 python bench.py "$@" --no-cpu-baseline 2>/dev/null | tail -1 > gpurun_out/${TAG}_bench_$tag.json
  rm -rf gpurun_out/prof_tmp
  MART_OVERLAP_WGRAD=0 MART_TWO_STREAM=0 timeout 600 rocprofv3 --kernel-trace --stats -d gpurun_out/prof_tmp -- python bench.py "$@" --steps 3 --warmup 1 --no-cpu-baseline --no-kernel-timing --train-only > /dev/null 2>&1
  DB=$(find gpurun_out/prof_tmp -name "*.db" | head -1)
  python tools/rocpd_stats.py $DB > gpurun_out/${TAG}_kernel_stats_${tag}_serial.csv
  rm -rf gpurun_out/prof_tmp
  python -c "import json; d=json.load(open('gpurun_out/${TAG}_bench_$tag.json')); print('$tag', d['value'], d['unit'], d['ms_per_step'], 'ms/step', (d.get('roofline') or {}).get('step_frac_of_mfma_peak'))"
}
run p49 --patch 32
run head2063 --entity-head 2063
run pretrain --task pretrain --seq-len 96
run pretrain_p49 --task pretrain --seq-len 96 --patch 32
run flava --model flava
run plain --weights plain
timeout 300 python tools/step_boundary.py 20 > gpurun_out/${TAG}_step_boundary.txt 2>&1
# un-profiled main-queue intervals (HIP events at phase boundaries) at both geometries
{ timeout 300 python tools/tail_marks.py --patch 16 --steps 12 2>&1 | grep -v amdgpu.ids; timeout 300 python tools/tail_marks.py --patch 32 --steps 12 2>&1 | grep -v amdgpu.ids; } > gpurun_out/${TAG}_tail_marks.txt
# queue timeline of the 49-patch geometry (shipped schedule) and the kernel-by-kernel listing of its tail
rm -rf gpurun_out/prof_tmp
timeout 600 rocprofv3 --kernel-trace --stats -d gpurun_out/prof_tmp -- python bench.py --patch 32 --steps 4 --warmup 2 --no-cpu-baseline --train-only --no-kernel-timing > /dev/null 2>&1
DB=$(find gpurun_out/prof_tmp -name "*.db" | head -1)
python tools/rocpd_streams.py $DB > gpurun_out/${TAG}_streams_overlap_p49.txt 2>&1
python tools/rocpd_tail.py $DB 700 1500 > gpurun_out/${TAG}_tail_p49.txt 2>&1
rm -rf gpurun_out/prof_tmp
python - <<PY
import json
d = json.load(open("gpurun_out/${TAG}_bench.json"))
print("BENCH", d["value"], d["ms_per_step"], "traffic", d["roofline"]["traffic"], d["roofline"]["traffic_source"])
PY
