#!/bin/bash
cd $GRAFT_REPO_ROOT
echo "== new"; timeout 300 python tools/bench_nt_cfg.py 0 2>&1 | grep "gemm_nt M"
cp mkg_analogy_amd/csrc/gemm_nt.hip /tmp/new_nt.hip; cp tools/_gemm_nt_head.hip.txt mkg_analogy_amd/csrc/gemm_nt.hip; python mkg_analogy_amd/_build.py 2>&1 | tail -1
echo "== committed"; timeout 300 python tools/bench_nt_cfg.py 0 2>&1 | grep "gemm_nt M"
cp /tmp/new_nt.hip mkg_analogy_amd/csrc/gemm_nt.hip; python mkg_analogy_amd/_build.py 2>&1 | tail -1
echo "== new again"; timeout 300 python tools/bench_nt_cfg.py 0 2>&1 | grep "gemm_nt M"
