#!/bin/bash
# One bundled GPU-box session: parity tests, bench line, rocprofv3 kernel trace and two PMC passes.
# Usage (from the repo root on the box): bash tools/gpu_round.sh <tag>
TAG=${1:-run}
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/$TAG
mkdir -p $OUT
export TMPDIR=/tmp
cd $R
( timeout 1200 python -m pytest tests -m gpu -x -q 2>&1 | tail -25; echo "rc=${PIPESTATUS[0]}" ) > $OUT/pytest_gpu.log
( timeout 300 python __graft_entry__.py smoke 2>&1 | tail -3 ) > $OUT/smoke.log
timeout 900 python bench.py > $OUT/bench.json 2> $OUT/bench.err; echo "rc=$?" >> $OUT/bench.err
cd /tmp
timeout 400 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/kt -o c2 -- python $R/bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-synth --no-exclusive > $OUT/kt.log 2>&1
timeout 400 rocprofv3 --pmc FETCH_SIZE --output-format csv -d $OUT/pmc_fetch -o c2 -- python $R/bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-synth --no-exclusive > $OUT/pmc_fetch.log 2>&1
timeout 400 rocprofv3 --pmc WRITE_SIZE --output-format csv -d $OUT/pmc_write -o c2 -- python $R/bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-synth --no-exclusive > $OUT/pmc_write.log 2>&1
cd $R
for d in pmc_fetch pmc_write; do
  f=$(find $OUT/$d -name '*counter_collection.csv' | head -1)
  [ -n "$f" ] && python tools/pmc_summary.py $f --md > $OUT/$d.md
done
find $OUT -name '*.csv' -size +8M -delete
ls -laR $OUT | head -60
tail -3 $OUT/pytest_gpu.log; cat $OUT/smoke.log; cat $OUT/bench.json | cut -c1-600
