#!/bin/bash
# usage: tools/ab_env.sh VAR val1 val2 ...   -- bench step time for each value of an environment variable, same box
cd $GRAFT_REPO_ROOT
var=$1; shift
run() { timeout 300 python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-kernel-timing 2>&1 | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('   ', d['value'], d['ms_per_step'], d['loss'])"; }
for v in "$@" "$1"; do echo "== $var=$v"; export $var=$v; run; run; done
