# usage (gpurun): bash tools/nt_aux.sh   -- cache-policy bits of the 8-phase loop's LDS-DMA, A/B in tools/nt_harness ab (base build vs each variant)
cd $GRAFT_REPO_ROOT
mkdir -p tools/variants
bash tools/build_variant.sh gemm_nt.hip tools/variants/libmart_hip.so -DMART_EXPERIMENTS > /dev/null 2>&1
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O2 -Iinclude tools/nt_harness.cpp -o tools/nt_harness -Ltools/variants -lmart_hip -Wl,-rpath,'$ORIGIN/variants' 2>&1 | grep -v warning | head -5
for v in "2 0" "0 2" "2 2" "1 0" "16 0" "17 0" "18 0" "0 1" "0 16"; do
  set -- $v
  bash tools/build_variant.sh gemm_nt.hip tools/variants/aux_$1_$2.so -DMART_EXPERIMENTS -DGLDS_AUX_A=$1 -DGLDS_AUX_B=$2 > /dev/null 2>&1 || { echo "build failed $v"; continue; }
  echo "== GLDS_AUX_A=$1 GLDS_AUX_B=$2"
  timeout 120 tools/nt_harness ab 5 tools/variants/libmart_hip.so tools/variants/aux_$1_$2.so 0 0 2>&1 | grep "^ab \|total" | awk "{print \$2, \$3, \$NF}" | tr "\n" ";"; echo
done
