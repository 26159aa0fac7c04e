#!/bin/bash
# usage (gpurun): bash tools/tn_one_transposed.sh  -- VERDICT r5 item 3, priced in the harness BEFORE touching any producer: the weight-gradient loop with the
# fragments of one operand (Y, X) or both from plain ds_read_b128 instead of transposed reads (knock-outs: wrong results, same traffic and MFMAs) = the UPPER
# bound of what an M-contiguous copy of that operand could buy; against the shipped kernel, alternating, three rounds
cd $GRAFT_REPO_ROOT
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O2 -Iinclude tools/tn_harness.cpp -o tools/tn_harness -Lmkg_analogy_amd/lib -lmart_hip -Wl,-rpath,'$ORIGIN/../mkg_analogy_amd/lib' 2>&1 | grep -v warning | head -3
for v in 0 1 2 3; do mkdir -p tools/variants/tn_ko$v; bash tools/build_variant.sh gemm_tn.hip tools/variants/tn_ko$v/libmart_hip.so -DMART_EXPERIMENTS -DTN_KO_PLAIN=$v > /dev/null 2>&1 || echo "build $v failed"; done
LD_LIBRARY_PATH=tools/variants/tn_ko0 timeout 200 tools/tn_harness check 2>&1 | tail -2
for r in 1 2 3; do for v in 0 1 2 3; do echo "== knock-out $v (0 shipped, 1 Y plain, 2 X plain, 3 both)"; LD_LIBRARY_PATH=tools/variants/tn_ko$v timeout 200 tools/tn_harness time 5 2>&1 | grep "^time" | cut -c1-170; done; done
