#!/bin/bash
cd $GRAFT_REPO_ROOT
echo "== opaque asm DMA"; timeout 300 python tools/bench_kernels.py 2>&1 | grep "attn_"
sed -i 's/glds16_raw(side_row/glds16(side_row/' mkg_analogy_amd/csrc/attention.hip
python mkg_analogy_amd/_build.py 2>&1 | tail -1
echo "== builtin DMA"; timeout 300 python tools/bench_kernels.py 2>&1 | grep "attn_"
