#!/bin/bash
# Evidence session: every number quoted in DESIGN.md / README.md that is not the headline gets a kept record.
#   bash tools/gpu_r2_evidence.sh <tag>
TAG=${1:-r2ev}
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/$TAG
mkdir -p $OUT
export TMPDIR=/tmp
cd $R
# whole GPU suite, verbose (-s): measured errors of every parity test
( timeout 1500 python -m pytest tests -m gpu -s -q 2>&1; echo "rc=$?" ) > $OUT/pytest_gpu_all_verbose.log
grep -c . $OUT/pytest_gpu_all_verbose.log; tail -3 $OUT/pytest_gpu_all_verbose.log
# other workloads through the same bench (one line each)
for w in default_hparams c2_4stack c5_stress; do
  timeout 300 python bench.py --workload $w --steps 10 --warmup 3 --no-cpu-baseline --no-synth --sustained 30 > $OUT/bench_$w.json 2> $OUT/bench_$w.err
  cut -c1-250 $OUT/bench_$w.json
done
( cd /tmp && timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/kt_c5 -o c5 -- python $R/bench.py --workload c5_stress --steps 3 --warmup 1 --no-cpu-baseline --no-synth --no-exclusive --sustained 0 > $OUT/kt_c5.log 2>&1 )
f=$(find $OUT/kt_c5 -name '*kernel_stats.csv' | head -1); [ -n "$f" ] && cp $f $OUT/c5_kernel_stats.csv; rm -rf $OUT/kt_c5
# hardware probes
timeout 120 tools/mfma_peak > $OUT/mfma_peak.txt 2>&1
timeout 120 tools/hop_probe > $OUT/hop_probe.txt 2>&1
# synthesis pipeline trace (per-stage latencies from s_memrealtime stamps) + kernel trace of one 5 s utterance
WN_PIPE_TRACE=1 timeout 200 python tools/pipe_trace.py 1 > $OUT/pipe_trace_b1.txt 2>&1
WN_PIPE_TRACE=1 timeout 200 python tools/pipe_trace.py 8 > $OUT/pipe_trace_b8.txt 2>&1
tail -4 $OUT/pipe_trace_b8.txt; tail -3 $OUT/mfma_peak.txt; tail -3 $OUT/hop_probe.txt
