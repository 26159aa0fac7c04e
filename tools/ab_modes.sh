#!/bin/bash
# usage: tools/ab_modes.sh "VAR=a VAR2=b" "VAR=c" ...   -- the bench step time for each environment setting, interleaved, twice, same box
cd $GRAFT_REPO_ROOT
run() { env $1 timeout 300 python bench.py --steps 12 --warmup 3 --no-cpu-baseline --no-kernel-timing --train-only 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('   ', d['value'], d['ms_per_step'], d['loss'])"; }
for rep in 1 2; do for v in "$@"; do echo "== [$v]"; run "$v"; done; done
