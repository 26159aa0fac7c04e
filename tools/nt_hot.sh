# usage (gpurun): bash tools/nt_hot.sh [harness args...]   -- builds the MART_EXPERIMENTS variant of gemm_nt + tools/nt_harness, then runs it (default: NT_HOT sweep)
cd $GRAFT_REPO_ROOT
mkdir -p tools/variants
bash tools/build_variant.sh gemm_nt.hip tools/variants/libmart_hip.so -DMART_EXPERIMENTS > /dev/null 2>&1
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O2 -Iinclude tools/nt_harness.cpp -o tools/nt_harness -Ltools/variants -lmart_hip -Wl,-rpath,'$ORIGIN/variants' 2>&1 | grep -v warning | head -5
if [ $# -gt 0 ]; then timeout 300 tools/nt_harness "$@" 2>&1 | grep -v "^$"; exit 0; fi
for h in 0 1 2; do echo "== NT_HOT=$h"; NT_HOT=$h timeout 120 tools/nt_harness time 5 0 0 2>&1 | grep -v "^$" | head -14; done
