#!/bin/bash
# wgrad harness + gradient parity + bench after a weight-gradient kernel change
TAG=${1:-r2p}
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/$TAG
mkdir -p $OUT
export TMPDIR=/tmp
cd $R
timeout 200 tools/wgrad_harness > $OUT/wgrad.txt 2>&1
( timeout 900 python -m pytest tests/test_hip_parity.py tests/test_hip_reference_golden.py -m gpu -x -q 2>&1 | tail -8; echo "rc=${PIPESTATUS[0]}" ) > $OUT/pytest_a.log
( timeout 900 python -m pytest tests/test_hip_bench_geometry.py -m gpu -s -q -k "b2_two or 4stack or buckets_are" 2>&1 | tail -30; echo "rc=${PIPESTATUS[0]}" ) > $OUT/pytest_b.log
B="--steps 40 --warmup 8 --no-cpu-baseline --no-synth --no-exclusive --sustained 0"
for v in "base:" "base:" "base:"; do
  name=${v%%:*}; envs=${v#*:}
  ( env $envs timeout 200 python bench.py $B 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$name', '%.3f ms/step' % d['ms_per_step'])" ) >> $OUT/ab.txt 2>&1
done
cat $OUT/ab.txt; tail -4 $OUT/pytest_a.log; tail -8 $OUT/pytest_b.log; tail -25 $OUT/wgrad.txt
