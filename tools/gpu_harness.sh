#!/bin/bash
# GEMM / wgrad harness tables + SQ PMC passes on the epilogue-ablated main loops.   bash tools/gpu_harness.sh <tag>
TAG=${1:-h}
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/$TAG
mkdir -p $OUT
export TMPDIR=/tmp
cd /tmp
rocprofv3 -L > $OUT/counters.txt 2>&1
timeout 120 $R/tools/gemm_harness > $OUT/gemm.txt 2>&1
timeout 120 $R/tools/gemm_harness_ablate > $OUT/gemm_ablate.txt 2>&1
timeout 120 $R/tools/wgrad_harness > $OUT/wgrad.txt 2>&1
i=0
for set in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_VALU_MFMA_BUSY_CYCLES" \
           "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_LDS SQ_ACTIVE_INST_VMEM SQ_INSTS_VALU_MFMA_MOPS_BF16 SQ_WAVES GRBM_GUI_ACTIVE" \
           "SQ_LDS_ADDR_CONFLICT SQ_LDS_UNALIGNED_STALL SQ_INST_CYCLES_VMEM SQ_ACTIVE_INST_MISC SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_VALU"; do
  i=$((i+1))
  timeout 200 rocprofv3 --pmc $set --output-format csv -d $OUT/pmc$i -o g -- $R/tools/gemm_harness_ablate > $OUT/pmc$i.log 2>&1
  f=$(find $OUT/pmc$i -name '*counter_collection.csv' | head -1)
  [ -n "$f" ] && python $R/tools/pmc_summary.py $f > $OUT/pmc$i.txt
  rm -rf $OUT/pmc$i
done
cat $OUT/gemm.txt $OUT/gemm_ablate.txt; tail -20 $OUT/wgrad.txt
