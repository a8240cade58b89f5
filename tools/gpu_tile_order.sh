#!/bin/bash
# One-shot A/B of the LDS-kernel tile order (WN_TILE_ORDER 1 = contiguous per XCD, 0 = interleaved): parity, step time, FETCH_SIZE.
TAG=${1:-order}
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/$TAG
mkdir -p $OUT
export TMPDIR=/tmp
cd $R
( WN_TILE_ORDER=1 timeout 200 python -m pytest tests -m gpu -x -q 2>&1 | tail -6; echo "rc=${PIPESTATUS[0]}" ) > $OUT/pytest_order1.log
B="--steps 20 --warmup 5 --no-cpu-baseline --no-synth --no-exclusive"
WN_TILE_ORDER=1 timeout 90 python bench.py $B > $OUT/bench_order1.json 2>/dev/null
WN_TILE_ORDER=0 timeout 90 python bench.py $B > $OUT/bench_order0.json 2>/dev/null
cd /tmp
for o in 1 0; do
  WN_TILE_ORDER=$o timeout 120 rocprofv3 --pmc FETCH_SIZE --output-format csv -d $OUT/pmc$o -o c2 -- python $R/bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-synth --no-exclusive > $OUT/pmc$o.log 2>&1
  f=$(find $OUT/pmc$o -name '*counter_collection.csv' | head -1)
  [ -n "$f" ] && python $R/tools/pmc_summary.py $f --md > $OUT/pmc_fetch_order$o.md
  rm -rf $OUT/pmc$o
done
cd $R
( WN_TILE_ORDER=0 timeout 100 python -m pytest tests/test_hip_reference_golden.py -m gpu -x -q 2>&1 | tail -3; echo "rc=${PIPESTATUS[0]}" ) > $OUT/pytest_order0.log
for o in 1 0; do python - <<PY
import json
try:
    d=json.loads(open('$OUT/bench_order$o.json').read().strip().splitlines()[-1]); print('order $o', round(d['ms_per_step'],3), 'ms', d['roofline']['avg_launch_ms'])
except Exception as e: print('order $o bench failed', e)
PY
done
tail -2 $OUT/pytest_order1.log; tail -2 $OUT/pytest_order0.log
grep "3, [05], 1>" $OUT/pmc_fetch_order1.md $OUT/pmc_fetch_order0.md
