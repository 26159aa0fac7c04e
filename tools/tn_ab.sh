# usage (gpurun): bash tools/tn_ab.sh "<hipcc flags of variant A>" ["<flags of variant B>"]  -- tools/tn_harness on two builds of gemm_tn.hip, alternating, twice
cd $GRAFT_REPO_ROOT
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O2 -Iinclude tools/tn_harness.cpp -o tools/tn_harness -Lmkg_analogy_amd/lib -lmart_hip -Wl,-rpath,'$ORIGIN/../mkg_analogy_amd/lib' 2>&1 | grep -v warning | head -3
mkdir -p tools/variants/tn_a tools/variants/tn_b
bash tools/build_variant.sh gemm_tn.hip tools/variants/tn_a/libmart_hip.so $1 > /dev/null 2>&1 || echo "build A failed"
bash tools/build_variant.sh gemm_tn.hip tools/variants/tn_b/libmart_hip.so $2 > /dev/null 2>&1 || echo "build B failed"
LD_LIBRARY_PATH=tools/variants/tn_a timeout 200 tools/tn_harness check 2>&1 | tail -3
for r in 1 2; do for v in a b; do echo "== variant $v"; LD_LIBRARY_PATH=tools/variants/tn_$v timeout 200 tools/tn_harness time 5 2>&1 | grep "^time" | cut -c1-150; done; done
