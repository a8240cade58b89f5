#!/bin/bash
# serial (exclusive) and live kernel statistics of the bench step
TAG=${1:-prof}
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/$TAG
mkdir -p $OUT
export TMPDIR=/tmp
cd /tmp
B="--steps 5 --warmup 2 --no-cpu-baseline --no-synth --no-exclusive --sustained 0"
WN_SERIAL=1 WN_BATCH_PARTS=1 timeout 400 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/serial -o c2 -- python $R/bench.py $B > $OUT/serial.log 2>&1
timeout 400 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/kt -o c2 -- python $R/bench.py $B > $OUT/kt.log 2>&1
cd $R
f=$(find $OUT/serial -name '*kernel_stats.csv' | head -1); [ -n "$f" ] && cp $f $OUT/serial_kernel_stats.csv
f=$(find $OUT/kt -name '*kernel_stats.csv' | head -1); [ -n "$f" ] && cp $f $OUT/kernel_stats.csv
f=$(find $OUT/kt -name '*kernel_trace.csv' | head -1); [ -n "$f" ] && python tools/timeline.py $f > $OUT/timeline.txt 2>&1
rm -rf $OUT/serial $OUT/kt
head -12 $OUT/timeline.txt
