#!/bin/bash
# Round 6, session b: whole-B kernel for the HBM-bound 1x1 convolutions -- harness (bit-identity + timing), step A/B, parity tests on it
R=${GRAFT_REPO_ROOT:-$(pwd)}; cd $R
TAG=r9b; OUT=$R/gpurun_out/$TAG; mkdir -p $OUT
( cd /tmp; timeout 300 $R/tools/stream_harness 8 11000 3 ) > $OUT/stream_harness.txt 2>&1; cat $OUT/stream_harness.txt
for i in 1 2; do
bash tools/gpu_session.sh $TAG "ab:wb0=WN_GEMM_WB=0"
bash tools/gpu_session.sh $TAG "ab:wb1=WN_GEMM_WB=1"
done
bash tools/gpu_session.sh $TAG "tests:test_hip_parity or test_hip_bench_geometry or test_hip_round3"
bash tools/gpu_session.sh $TAG benchw:default_hparams
WN_GEMM_WB=0 timeout 300 python bench.py --workload default_hparams --steps 20 --warmup 5 --no-cpu-baseline --no-synth --no-exclusive --no-other-workloads --sustained 0 2>/dev/null | cut -c1-200
