#!/bin/bash
# One GPU-box session for the synthesis pipeline:  bash tools/gpu_pipe_session.sh <tag> [tests-expr] [svc B ...]
#   parity tests matching the expression, per-sample time against the number of streams (tools/pipe_batch_scaling.py), and -- diagnostic build --
#   where a layer CU's service time per stream goes (tools/pipe_svc_trace.py)
TAG=${1:-p}; EXPR=${2:-pipe}; shift; shift
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/$TAG
mkdir -p $OUT
cd $R
( timeout 900 python -m pytest tests -m gpu -q -s --maxfail=6 -k "$EXPR" 2>&1; echo "rc=$?" ) > $OUT/pytest_pipe.log; grep -E "streams|passed|failed|error|rc=" $OUT/pytest_pipe.log | tail -30
timeout 600 python tools/pipe_batch_scaling.py 1.0 > $OUT/pipe_batch_scaling.json 2> $OUT/pipe_batch_scaling.txt; cat $OUT/pipe_batch_scaling.txt | cut -c1-400
for v in ${PIPE_AB:-}; do     # A/B arms of the c2 scaling table: PIPE_AB="WN_PIPE_EARLY_FROM=99 WN_PIPE_BATCHPRE=0"
  echo "--- $v" >> $OUT/pipe_batch_scaling_ab.txt
  env $v timeout 300 python tools/pipe_batch_scaling.py 1.0 c2 8,12,16,20,24 2>> $OUT/pipe_batch_scaling_ab.txt > /dev/null
done
[ -f $OUT/pipe_batch_scaling_ab.txt ] && grep -v amdgpu.ids $OUT/pipe_batch_scaling_ab.txt | cut -c1-300
if [ -n "$1" ]; then
  python tacotron-2_amd/csrc/build.py --pipe-svc > /dev/null 2>&1
  timeout 300 python tools/pipe_svc_trace.py "$@" > $OUT/pipe_svc_trace.txt 2>&1; cat $OUT/pipe_svc_trace.txt
  WN_PIPE_BATCHPRE=0 timeout 300 python tools/pipe_svc_trace.py "$@" > $OUT/pipe_svc_trace_per_stream_form.txt 2>&1; cat $OUT/pipe_svc_trace_per_stream_form.txt
fi
