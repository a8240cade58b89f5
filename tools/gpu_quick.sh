#!/bin/bash
# parity (gradients incl. the upsample net) + 3 bench runs + serial kernel stats
TAG=${1:-quick}
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/$TAG
mkdir -p $OUT
cd $R
( timeout 900 python -m pytest tests/test_hip_parity.py tests/test_hip_reference_golden.py -m gpu -x -q 2>&1 | tail -4; echo "rc=${PIPESTATUS[0]}" ) > $OUT/pytest_a.log
tail -3 $OUT/pytest_a.log
bash tools/gpu_ab.sh $TAG "base:" "base:" "base:"
export TMPDIR=/tmp
cd /tmp
B="--steps 5 --warmup 2 --no-cpu-baseline --no-synth --no-exclusive --sustained 0"
WN_SERIAL=1 WN_BATCH_PARTS=1 timeout 400 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/serial -o c2 -- python $R/bench.py $B > $OUT/serial.log 2>&1
cd $R
f=$(find $OUT/serial -name '*kernel_stats.csv' | head -1); [ -n "$f" ] && cp $f $OUT/serial_kernel_stats.csv
rm -rf $OUT/serial
python - <<P
import csv
rows=list(csv.DictReader(open('$OUT/serial_kernel_stats.csv')))
tot=sum(float(r['TotalDurationNs']) for r in rows)
print('serial total kernel ms per step', tot/7/1e6)
for r in sorted(rows,key=lambda r:-float(r['TotalDurationNs']))[:12]:
    print('  %-74s %5s %9.1f us avg  per-step %.0f us'%(r['Name'][:74],r['Calls'],float(r['AverageNs'])/1e3, float(r['TotalDurationNs'])/7e3))
P
