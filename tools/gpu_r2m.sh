#!/bin/bash
# harness + short parity + A/B bench after an epilogue change
TAG=${1:-r2m}
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/$TAG
mkdir -p $OUT
export TMPDIR=/tmp
cd $R
timeout 200 tools/gemm_harness 8 > $OUT/gemm_b8.txt 2>&1
( timeout 900 python -m pytest tests/test_hip_parity.py tests/test_hip_reference_golden.py tests/test_hip_synth.py -m gpu -x -q 2>&1 | tail -8; echo "rc=${PIPESTATUS[0]}" ) > $OUT/pytest_a.log
( timeout 900 python -m pytest tests/test_hip_bench_geometry.py -m gpu -s -q -k "rccl or b2_two" 2>&1 | tail -40; echo "rc=${PIPESTATUS[0]}" ) > $OUT/pytest_b.log
B="--steps 40 --warmup 8 --no-cpu-baseline --no-synth --no-exclusive --sustained 0"
for v in "base:" "base:" "base:"; do
  name=${v%%:*}; envs=${v#*:}
  ( env $envs timeout 200 python bench.py $B 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$name', '%.3f ms/step' % d['ms_per_step'])" ) >> $OUT/ab.txt 2>&1
done
cat $OUT/ab.txt; tail -4 $OUT/pytest_a.log; tail -14 $OUT/pytest_b.log
grep -E "^(gate|out|dx|dgate|skip) " $OUT/gemm_b8.txt | head -40
