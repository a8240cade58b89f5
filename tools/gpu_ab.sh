#!/bin/bash
# bench-only A/B sweep: tools/gpu_ab.sh <tag> "name:ENV=1 ENV2=2" ...
TAG=$1; shift
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/$TAG
mkdir -p $OUT
cd $R
B="--steps 40 --warmup 8 --no-cpu-baseline --no-synth --no-exclusive --sustained 0"
for v in "$@"; do
  name=${v%%:*}; envs=${v#*:}
  ( env $envs timeout 200 python bench.py $B 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$name', '%.3f ms/step' % d['ms_per_step'])" ) >> $OUT/ab.txt 2>&1
done
cat $OUT/ab.txt
