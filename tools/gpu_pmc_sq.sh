#!/bin/bash
# SQ-level PMC pass over the training step (single-stream order so that counters attribute to one kernel at a time)
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/${1:-sq}
mkdir -p $OUT
export TMPDIR=/tmp
cd /tmp
WN_BATCH_PARTS=1 timeout 300 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE GRBM_GUI_ACTIVE \
  --output-format csv -d $OUT/pmc -o c2 -- python $R/bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-synth --no-exclusive > $OUT/pmc.log 2>&1
f=$(find $OUT/pmc -name '*counter_collection.csv' | head -1)
[ -n "$f" ] && python $R/tools/pmc_summary.py $f > $OUT/pmc_sq.txt
rm -rf $OUT/pmc
tail -2 $OUT/pmc.log | cut -c1-200; head -30 $OUT/pmc_sq.txt | cut -c1-60,112-
