#!/bin/bash
# Round 6, session d: harnesses on the product headers with energy per launch; 8-phase kernel with the prologue wait removed (bound of a persistent grid);
# geometries with an even tile count (what perfect balance would give); default-model profile (kernel stats + PMC traffic)
R=${GRAFT_REPO_ROOT:-$(pwd)}; cd $R
TAG=r9d; OUT=$R/gpurun_out/$TAG; mkdir -p $OUT
( cd /tmp; timeout 400 $R/tools/gemm8p_harness 8 11000 64 2 ) > $OUT/gemm8p_harness.txt 2>&1; grep -E "energy|prologue|production|FAIL|passed|wider" $OUT/gemm8p_harness.txt | head -40
( cd /tmp; timeout 300 $R/tools/gemm8p_harness 8 8192 64 2 ) > $OUT/gemm8p_harness_even_tiles_8x8192.txt 2>&1; grep -E "production|prologue" $OUT/gemm8p_harness_even_tiles_8x8192.txt | head
( cd /tmp; timeout 300 $R/tools/stream_harness 8 11000 2 ) > $OUT/stream_harness.txt 2>&1; grep -E "energy|d z" $OUT/stream_harness.txt
bash tools/gpu_session.sh $TAG ktw:default_hparams
cd /tmp
for c in FETCH_SIZE WRITE_SIZE; do
  n=$(echo $c | tr 'A-Z' 'a-z' | sed 's/_size//')
  timeout 400 rocprofv3 --pmc $c --output-format csv -d $OUT/pmc_$n -o w -- python $R/bench.py --workload default_hparams --steps 2 --warmup 1 --no-cpu-baseline --no-synth --no-exclusive --no-other-workloads --sustained 0 > $OUT/pmc_default_$n.log 2>&1
  f=$(find $OUT/pmc_$n -name '*counter_collection.csv' | head -1); [ -n "$f" ] && python $R/tools/pmc_summary.py $f --md > $OUT/pmc_default_hparams_$n.md
  rm -rf $OUT/pmc_$n
done
head -12 $OUT/pmc_default_hparams_fetch.md; head -12 $OUT/pmc_default_hparams_write.md
