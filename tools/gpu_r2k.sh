#!/bin/bash
# Focused session: upsample-net backward rewrite (parity + A/B), RCCL single-rank bucket walk, serial (exclusive) kernel profile.
TAG=${1:-r2k}
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/$TAG
mkdir -p $OUT
export TMPDIR=/tmp
cd $R
( timeout 900 python -m pytest tests/test_hip_parity.py tests/test_hip_reference_golden.py tests/test_hip_drivers.py -m gpu -x -q 2>&1 | tail -15; echo "rc=${PIPESTATUS[0]}" ) > $OUT/pytest_a.log
( timeout 900 python -m pytest tests/test_hip_bench_geometry.py -m gpu -s -q -k "rccl or b2_two or c5_width" 2>&1 | tail -60; echo "rc=${PIPESTATUS[0]}" ) > $OUT/pytest_b.log
B="--steps 40 --warmup 8 --no-cpu-baseline --no-synth --no-exclusive --sustained 0"
for v in "base:" "up_v1:WN_UP_BWD_V1=1" "base:" "up_v1:WN_UP_BWD_V1=1" "b3:WN_BWD_BUCKETS=3" "b3_up_v1:WN_BWD_BUCKETS=3 WN_UP_BWD_V1=1"; do
  name=${v%%:*}; envs=${v#*:}
  ( env $envs timeout 200 python bench.py $B 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$name', '%.3f ms/step' % d['ms_per_step'])" ) >> $OUT/ab.txt 2>&1
done
cat $OUT/ab.txt
cd /tmp
WN_SERIAL=1 WN_BATCH_PARTS=1 timeout 400 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/serial -o c2 -- python $R/bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-synth --no-exclusive --sustained 0 > $OUT/serial.log 2>&1
timeout 400 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/kt -o c2 -- python $R/bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-synth --no-exclusive --sustained 0 > $OUT/kt.log 2>&1
cd $R
f=$(find $OUT/serial -name '*kernel_stats.csv' | head -1); [ -n "$f" ] && cp $f $OUT/serial_kernel_stats.csv
f=$(find $OUT/kt -name '*kernel_stats.csv' | head -1); [ -n "$f" ] && cp $f $OUT/kernel_stats.csv
f=$(find $OUT/kt -name '*kernel_trace.csv' | head -1); [ -n "$f" ] && python tools/timeline.py $f > $OUT/timeline.txt 2>&1
find $OUT -name '*.csv' -size +8M -delete
tail -4 $OUT/pytest_a.log; tail -12 $OUT/pytest_b.log; tail -2 $OUT/serial.log | cut -c1-200
