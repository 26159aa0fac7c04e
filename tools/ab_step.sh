#!/bin/bash
cd $GRAFT_REPO_ROOT
run() { timeout 300 python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-kernel-timing 2>&1 | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('   ', d['value'], d['ms_per_step'])"; }
echo "== new gemm_tn"; run; run
cp mkg_analogy_amd/csrc/gemm_tn.hip /tmp/new_tn.hip; cp tools/_gemm_tn_prev.hip.txt mkg_analogy_amd/csrc/gemm_tn.hip; python mkg_analogy_amd/_build.py 2>&1 | tail -1
echo "== previous gemm_tn (MFMA colsum)"; run; run
cp /tmp/new_tn.hip mkg_analogy_amd/csrc/gemm_tn.hip; python mkg_analogy_amd/_build.py 2>&1 | tail -1
echo "== new again"; run
echo "== new, overlap off"; MART_OVERLAP_WGRAD=0 run
