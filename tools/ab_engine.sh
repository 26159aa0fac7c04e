#!/bin/bash
cd $GRAFT_REPO_ROOT
run() { timeout 300 python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-kernel-timing 2>&1 | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('   ', d['value'], d['ms_per_step'], d['loss'])"; }
echo "== new engine"; run; run
cp mkg_analogy_amd/engine.py /tmp/engine_new.py; cp tools/_engine_prev.py.txt mkg_analogy_amd/engine.py
echo "== previous engine"; run; run
cp /tmp/engine_new.py mkg_analogy_amd/engine.py
echo "== new engine"; run
