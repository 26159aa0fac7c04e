#!/bin/bash
# usage (gpurun): bash tools/nt_store_policy.sh  -- cache policy of gemm_nt's epilogue stores (where do the extra fabric bytes come from, and do
# the output lines have to sit in L2 at all?): the shipped nt stores against sc1 (write-through + drop from L2), sc0 sc1, sc1 nt; tools/nt_harness ab, five rounds
cd $GRAFT_REPO_ROOT
mkdir -p tools/variants
bash tools/build_variant.sh gemm_nt.hip tools/variants/libmart_hip.so -DMART_EXPERIMENTS > /dev/null 2>&1
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O2 -Iinclude tools/nt_harness.cpp -o tools/nt_harness -Ltools/variants -lmart_hip -Wl,-rpath,'$ORIGIN/variants' 2>&1 | grep -v warning | head -5
for v in 1 2 3; do
  bash tools/build_variant.sh gemm_nt.hip tools/variants/st_policy_$v.so -DMART_EXPERIMENTS -DMART_ST_POLICY=$v > /dev/null 2>&1 || { echo "build failed $v"; continue; }
  echo "== MART_ST_POLICY=$v (1 sc1, 2 sc0 sc1, 3 sc1 nt)  A = shipped (nt), B = variant"
  timeout 200 tools/nt_harness ab 5 tools/variants/libmart_hip.so tools/variants/st_policy_$v.so 0 0 2>&1 | grep "^ab \|total" | cut -c1-200
done
