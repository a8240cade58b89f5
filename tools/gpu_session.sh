#!/bin/bash
# One GPU-box session, parameterised (replaces the one-off gpu_r2*.sh scripts of round 2).
#   bash tools/gpu_session.sh <tag> <stage> [<stage> ...]
# stages:
#   tests        full `pytest -m gpu` suite, verbose (-s keeps the measured distances), junit-free log
#   tests:<expr> only the tests matching -k <expr>
#   smoke        __graft_entry__.smoke()
#   bench        default-flag bench line (what the driver runs) -> bench.json
#   forcedist    short bench through the one-rank RCCL path (process group, bucketed ReduceOp.AVG exchange, barriers) -> bench_forcedist.json
#   benchq       short bench (no synth / cpu baseline / other workloads) -> benchq.json
#   ab:<name>=<ENV=V>[,<ENV=V>]   one A/B arm of the short bench (appends to ab.txt); `ab:base=` for the baseline
#   kt           rocprofv3 kernel trace + stats of the bench step (live, two streams) -> kernel_stats.csv, timeline.txt
#   roctx        WN_ROCTX=1: rocTX ranges of the C-ABI entry points in a rocprofv3 marker + kernel trace -> roctx_marker_stats.csv
#   serial       the same with every kernel alone on one stream (exclusive kernel times) -> serial_kernel_stats.csv
#   pmc          FETCH_SIZE and WRITE_SIZE passes -> pmc_fetch.md, pmc_write.md, traffic.json (bench.py loads the committed copy)
#   sq           SQ counter passes of the bench step -> pmc_sq.md
#   harness      the stand-alone harnesses on the PRODUCT headers: gemm8p_harness (gate / d x: ring vs 8-phase), stream_harness (out conv, mask GEMM, d z), wgrad_harness
#   chain        tools/chain_harness: persistent layer-chain prototype vs the two-stream schedule (forward, 6 layers) -> chain_harness.txt
#   pipetrace    WN_PIPE_TRACE stage trace of the synthesis pipeline (B = 1, 8) -> pipe_trace_b*.txt
#   other        10-step runs of the other workloads only
#   devtrace     the engine's own in-kernel stamps of one un-profiled step (WN_DEVTRACE) -> devtrace.txt, devtrace_timeline.txt
#   benchw:<w>   short bench of workload <w> (c2_4stack, default_hparams, c5_stress); ktw:<w> its rocprofv3 kernel statistics
TAG=${1:-s}; shift
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/$TAG
mkdir -p $OUT
export TMPDIR=/tmp
export WN_PARITY_REPORT_DIR=$OUT
BQ="--no-cpu-baseline --no-synth --no-exclusive --no-other-workloads --sustained 0"
for st in "$@"; do
  cd $R
  case $st in
    tests) ( timeout 1500 python -m pytest tests -m gpu -q -s --maxfail=8 2>&1; echo "rc=$?" ) > $OUT/pytest_gpu_verbose.log; grep -E "passed|failed|error|rc=" $OUT/pytest_gpu_verbose.log | tail -5 ;;
    tests:*) ( timeout 1200 python -m pytest tests -m gpu -q -s --maxfail=8 -k "${st#tests:}" 2>&1; echo "rc=$?" ) > $OUT/pytest_gpu_k.log; grep -E "passed|failed|error|rc=" $OUT/pytest_gpu_k.log | tail -5 ;;
    smoke) ( timeout 300 python __graft_entry__.py smoke 2>&1 | tail -3 ) > $OUT/smoke.log; cat $OUT/smoke.log ;;
    bench) timeout 900 python bench.py > $OUT/bench.json 2> $OUT/bench.err; echo "rc=$?" >> $OUT/bench.err; cut -c1-400 $OUT/bench.json; tail -3 $OUT/bench.err ;;
    forcedist) GPU_MAX_HW_QUEUES=8 timeout 300 python bench.py --force-dist --steps 40 --warmup 8 $BQ > $OUT/bench_forcedist.json 2> $OUT/bench_forcedist.err; cut -c1-300 $OUT/bench_forcedist.json; grep -i "process group" $OUT/bench_forcedist.err | tail -2 ;;
    benchq) timeout 300 python bench.py --steps 40 --warmup 8 $BQ > $OUT/benchq.json 2> $OUT/benchq.err; cut -c1-300 $OUT/benchq.json ;;
    ab:*) spec=${st#ab:}; name=${spec%%=*}; envs=$(echo "${spec#*=}" | tr ',' ' ')
      ( env $envs timeout 240 python bench.py --steps 40 --warmup 8 $BQ 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$name', '%.3f ms/step' % d['ms_per_step'], 'gate frac %.3f (incl. wait %.3f)' % (d['roofline']['frac'], d['roofline']['frac_incl_queue_wait']))" ) >> $OUT/ab.txt 2>&1; tail -1 $OUT/ab.txt ;;
    kt) cd /tmp; timeout 400 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/kt -o c2 -- python $R/bench.py --steps 5 --warmup 2 $BQ > $OUT/kt.log 2>&1; cd $R
      f=$(find $OUT/kt -name '*kernel_stats.csv' | head -1); [ -n "$f" ] && cp $f $OUT/kernel_stats.csv
      f=$(find $OUT/kt -name '*kernel_trace.csv' | head -1); [ -n "$f" ] && python tools/timeline.py $f 1 --detail > $OUT/timeline.txt 2>&1
      rm -rf $OUT/kt; head -14 $OUT/timeline.txt ;;
    roctx) cd /tmp; WN_ROCTX=1 timeout 400 rocprofv3 --marker-trace --kernel-trace --stats --output-format csv -d $OUT/roctx -o c2 -- python $R/bench.py --steps 5 --warmup 2 $BQ --no-feeder > $OUT/roctx.log 2>&1; cd $R
      f=$(find $OUT/roctx -name '*marker_api_stats.csv' | head -1); [ -n "$f" ] && cp $f $OUT/roctx_marker_stats.csv
      f=$(find $OUT/roctx -name '*marker_api_trace.csv' | head -1); [ -n "$f" ] && head -60 $f > $OUT/roctx_marker_trace_head.csv
      rm -rf $OUT/roctx; cat $OUT/roctx_marker_stats.csv 2>/dev/null | cut -c1-160 ;;
    ktw:*) w=${st#ktw:}; cd /tmp; timeout 400 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/ktw -o w -- python $R/bench.py --workload $w --steps 10 --warmup 5 $BQ > $OUT/ktw_$w.log 2>&1; cd $R
      f=$(find $OUT/ktw -name '*kernel_stats.csv' | head -1); [ -n "$f" ] && cp $f $OUT/kernel_stats_$w.csv
      rm -rf $OUT/ktw; tail -2 $OUT/ktw_$w.log | cut -c1-200; head -16 $OUT/kernel_stats_$w.csv | cut -c1-160 ;;
    benchw:*) w=${st#benchw:}; timeout 300 python bench.py --workload $w --steps 20 --warmup 5 $BQ > $OUT/bench_$w.json 2> $OUT/bench_$w.err; cut -c1-200 $OUT/bench_$w.json ;;
    serial) cd /tmp; WN_SERIAL=1 WN_BATCH_PARTS=1 timeout 400 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/serial -o c2 -- python $R/bench.py --steps 5 --warmup 2 $BQ > $OUT/serial.log 2>&1; cd $R
      f=$(find $OUT/serial -name '*kernel_stats.csv' | head -1); [ -n "$f" ] && cp $f $OUT/serial_kernel_stats.csv
      rm -rf $OUT/serial; head -12 $OUT/serial_kernel_stats.csv | cut -c1-150 ;;
    pmc) cd /tmp
      for c in FETCH_SIZE WRITE_SIZE; do
        n=$(echo $c | tr 'A-Z' 'a-z' | sed 's/_size//')
        timeout 400 rocprofv3 --pmc $c --output-format csv -d $OUT/pmc_$n -o c2 -- python $R/bench.py --steps 2 --warmup 1 $BQ > $OUT/pmc_$n.log 2>&1
        f=$(find $OUT/pmc_$n -name '*counter_collection.csv' | head -1); [ -n "$f" ] && cp $f $OUT/pmc_$n.csv && python $R/tools/pmc_summary.py $f --md > $OUT/pmc_$n.md
        rm -rf $OUT/pmc_$n
      done
      cd $R; python tools/pmc_summary.py --traffic $OUT/pmc_fetch.csv $OUT/pmc_write.csv --tag $TAG --fetch-name profiles/${TAG}_pmc_fetch.md --write-name profiles/${TAG}_pmc_write.md --out $OUT/traffic.json
      rm -f $OUT/pmc_fetch.csv $OUT/pmc_write.csv ;;
    sq) cd /tmp; i=0
      for set in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_VALU_MFMA_BUSY_CYCLES"; do
        i=$((i+1)); timeout 400 rocprofv3 --pmc $set --output-format csv -d $OUT/sq$i -o c2 -- python $R/bench.py --steps 2 --warmup 1 $BQ > $OUT/sq$i.log 2>&1
        f=$(find $OUT/sq$i -name '*counter_collection.csv' | head -1); [ -n "$f" ] && python $R/tools/pmc_summary.py $f --md > $OUT/pmc_sq$i.md
        rm -rf $OUT/sq$i
      done; cd $R ;;
    harness) cd /tmp
      for h in gemm8p_harness stream_harness wgrad_harness; do [ -x $R/tools/$h ] && timeout 240 $R/tools/$h > $OUT/$h.txt 2>&1; done
      cd $R; tail -30 $OUT/gemm8p_harness.txt; tail -20 $OUT/stream_harness.txt ;;
    chain) cd /tmp; [ -x $R/tools/chain_harness ] && timeout 90 $R/tools/chain_harness > $OUT/chain_harness.txt 2>&1; cd $R; tail -16 $OUT/chain_harness.txt ;;
    chaindbg:*) cd /tmp; for f in $(echo ${st#chaindbg:} | tr ',' ' '); do timeout 60 $R/tools/chain_harness $f 2>&1 | grep -v "^    wg" | head -30 > $OUT/chain_dbg_$f.txt; echo "== flags $f"; head -12 $OUT/chain_dbg_$f.txt; done; cd $R ;;
    pipetrace) python tacotron-2_amd/csrc/build.py --pipe-svc > /dev/null 2>&1      # (the stamp sites are a diagnostic build; the box is discarded after the session)
      for b in 1 8; do WN_PIPE_TRACE=1 timeout 200 python tools/pipe_trace.py $b > $OUT/pipe_trace_b$b.txt 2>&1; tail -4 $OUT/pipe_trace_b$b.txt; done ;;
    other) timeout 900 python - > $OUT/other_workloads.json 2> $OUT/other.err <<'PY'
import json, sys, os
sys.path.insert(0, os.environ.get('GRAFT_REPO_ROOT', '.'))
import bench, torch
print(json.dumps({k: bench.other_workload_subprocess(k, 0) for k in ('default_hparams', 'c2_4stack', 'c5_stress', 'c2_fp32')}, indent=1))      # a fresh process each (HW queue assignment)
PY
      cut -c1-600 $OUT/other_workloads.json ;;
    devtrace) WN_DEVTRACE=$OUT/devtrace.txt timeout 300 python bench.py --steps 12 --warmup 3 $BQ > $OUT/devtrace_bench.json 2> $OUT/devtrace.err
      python tools/devtrace.py $OUT/devtrace.txt --all > $OUT/devtrace_timeline.txt 2>&1; head -30 $OUT/devtrace_timeline.txt ;;
    *) echo "unknown stage $st" ;;
  esac
done
find $OUT -name '*.csv' -size +8M -delete
ls $OUT
