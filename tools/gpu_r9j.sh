#!/bin/bash
# Round 6, session j: epilogue operand loads a pass ahead -- bit-identity (harness), phase stamps, step A/B against the previous library, parity subset
R=${GRAFT_REPO_ROOT:-$(pwd)}; cd $R
TAG=r9j; OUT=$R/gpurun_out/$TAG; mkdir -p $OUT
( cd /tmp; timeout 300 $R/tools/stream_harness_phases 8 11000 1 ) > $OUT/stream_harness_phases.txt 2>&1; grep -E "^phases|bitwise|FAIL|passed" $OUT/stream_harness_phases.txt | cut -c1-320
( cd /tmp; timeout 300 $R/tools/gemm8p_harness 8 11000 64 1 ) > $OUT/gemm8p_harness.txt 2>&1; grep -E "vs production|FAIL|passed" $OUT/gemm8p_harness.txt | cut -c1-200
PREV=$R/tacotron-2_amd/csrc/libwavenet_mi355_prev.so
for i in 1 2 3; do
bash tools/gpu_session.sh $TAG "ab:prev=WN_MI355_TEST_LIB=$PREV"
bash tools/gpu_session.sh $TAG "ab:new="
done
bash tools/gpu_session.sh $TAG "tests:test_hip_parity or test_hip_bench_geometry or test_hip_round3 or reproducible"
WN_MI355_TEST_LIB=$PREV timeout 300 python bench.py --workload default_hparams --steps 20 --warmup 5 --no-cpu-baseline --no-synth --no-exclusive --no-other-workloads --sustained 0 2>/dev/null | cut -c1-160
timeout 300 python bench.py --workload default_hparams --steps 20 --warmup 5 --no-cpu-baseline --no-synth --no-exclusive --no-other-workloads --sustained 0 2>/dev/null | cut -c1-160
WN_MI355_TEST_LIB=$PREV timeout 300 python bench.py --workload c5_stress --steps 10 --warmup 3 --no-cpu-baseline --no-synth --no-exclusive --no-other-workloads --sustained 0 2>/dev/null | cut -c1-160
timeout 300 python bench.py --workload c5_stress --steps 10 --warmup 3 --no-cpu-baseline --no-synth --no-exclusive --no-other-workloads --sustained 0 2>/dev/null | cut -c1-160
