#!/bin/bash
cd $GRAFT_REPO_ROOT
echo "== opaque asm DMA"; timeout 300 python tools/bench_tn.py 2>&1 | grep "splits=0"
sed -i 's/glds16_raw(/glds16(/' mkg_analogy_amd/csrc/gemm_tn.hip
python mkg_analogy_amd/_build.py 2>&1 | tail -1
echo "== builtin DMA"; timeout 300 python tools/bench_tn.py 2>&1 | grep "splits=0"
