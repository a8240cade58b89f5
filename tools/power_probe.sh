#!/bin/bash
# sample rocm-smi power / clocks while the bench loop runs (is the training step power-limited?)
R=${GRAFT_REPO_ROOT:-$(pwd)}
mkdir -p $R/gpurun_out
cd $R
rocm-smi --showmaxpower --showpower --showclocks 2>&1 | grep -v "^$" | head -30 > gpurun_out/power_idle.txt
python bench.py --no-cpu-baseline --no-synth --steps 600 --warmup 5 > gpurun_out/power_bench.json 2>/dev/null &
BP=$!
sleep 4
for i in 1 2 3 4 5 6; do rocm-smi --showpower --showclocks 2>&1 | grep -i "power\|sclk\|mclk" ; sleep 0.7; done > gpurun_out/power_load.txt
wait $BP
cat gpurun_out/power_idle.txt | head -20; echo ----; cat gpurun_out/power_load.txt | head -40; cut -c1-200 gpurun_out/power_bench.json
