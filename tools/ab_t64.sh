#!/bin/bash
# text attention backward (fused 64 x 64 kernel): kernel time under rocprofv3 for the shipped build and variants
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
for l in "" "$@"; do
  rm -rf /tmp/pt64; (cd /tmp && MART_HIP_LIB=${l:+$GRAFT_REPO_ROOT/$l} rocprofv3 --kernel-trace --stats -d /tmp/pt64 -o t -- python $GRAFT_REPO_ROOT/tools/bench_attn_text.py > /dev/null 2>&1)
  echo "== ${l:-shipped} P=${P:-0.1}"; python tools/rocpd_stats.py /tmp/pt64/t_results.db | grep "attn_" | cut -d, -f1,2,4 | sed 's/_ZN12_GLOBAL__N_1//'
done
