#!/bin/bash
# One bundled GPU-box session of round 2: full GPU suite, verbose geometry parity, bench line, rocprofv3 kernel trace, PMC passes.
# Usage (from the repo root on the box): bash tools/gpu_r2.sh <tag> [quick]
TAG=${1:-r2}
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/$TAG
mkdir -p $OUT
export TMPDIR=/tmp
export WN_PARITY_REPORT_DIR=$OUT
cd $R
( timeout 900 python -m pytest tests -m gpu -x -q --deselect tests/test_hip_bench_geometry.py -k "not c4_scale" 2>&1 | tail -15; echo "rc=${PIPESTATUS[0]}" ) > $OUT/pytest_gpu.log
( timeout 1200 python -m pytest tests/test_hip_bench_geometry.py tests/test_hip_synth_pipe.py tests/test_hip_boundary.py -m gpu -s -q -k "geometry or c4_scale or c5 or buckets or boundary" 2>&1; echo "rc=$?" ) > $OUT/pytest_gpu_verbose.log
( timeout 300 python __graft_entry__.py smoke 2>&1 | tail -3 ) > $OUT/smoke.log
timeout 600 python bench.py --steps 20 --warmup 5 --no-cpu-baseline > $OUT/bench.json 2> $OUT/bench.err; echo "rc=$?" >> $OUT/bench.err
# A/B switches (same box, same process structure): multi-A weight-gradient workgroups, gradient buckets, batch parts
for v in "base:" "wgrad1:WN_WGRAD_MULTI=0" "buckets3:WN_BWD_BUCKETS=3" "parts1:WN_BATCH_PARTS=1" "bigtiles:WN_SMALL_TILES=0" "base:"; do
  name=${v%%:*}; envs=${v#*:}
  ( env $envs timeout 200 python bench.py --steps 40 --warmup 8 --no-cpu-baseline --no-synth --no-exclusive --sustained 0 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$name', '%.3f ms/step' % d['ms_per_step'], 'gate frac %.3f' % d['roofline']['frac'])" ) >> $OUT/ab.txt 2>&1
done
cat $OUT/ab.txt
if [ "$2" != "quick" ]; then
cd /tmp
timeout 400 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/kt -o c2 -- python $R/bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-synth --no-exclusive --sustained 0 > $OUT/kt.log 2>&1
timeout 400 rocprofv3 --pmc FETCH_SIZE --output-format csv -d $OUT/pmc_fetch -o c2 -- python $R/bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-synth --no-exclusive --sustained 0 > $OUT/pmc_fetch.log 2>&1
timeout 400 rocprofv3 --pmc WRITE_SIZE --output-format csv -d $OUT/pmc_write -o c2 -- python $R/bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-synth --no-exclusive --sustained 0 > $OUT/pmc_write.log 2>&1
cd $R
for d in pmc_fetch pmc_write; do
  f=$(find $OUT/$d -name '*counter_collection.csv' | head -1)
  [ -n "$f" ] && python tools/pmc_summary.py $f --md > $OUT/$d.md
done
f=$(find $OUT/kt -name '*kernel_stats.csv' | head -1); [ -n "$f" ] && cp $f $OUT/kernel_stats.csv
f=$(find $OUT/kt -name '*kernel_trace.csv' | head -1); [ -n "$f" ] && python tools/timeline.py $f > $OUT/timeline.txt 2>&1
find $OUT -name '*.csv' -size +8M -delete
fi
tail -3 $OUT/pytest_gpu.log; grep -c . $OUT/pytest_gpu_verbose.log; tail -4 $OUT/pytest_gpu_verbose.log; cat $OUT/smoke.log; cut -c1-300 $OUT/bench.json; tail -2 $OUT/bench.err
