#!/bin/bash
# Quick pipeline timing session:  bash tools/gpu_pipe_quick.sh <tag> <B,B,...> [svc B]   (c2 model; samples the shader clock while the pipeline runs)
TAG=${1:-q}; BS=${2:-8,12,16,20,24}
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/$TAG
mkdir -p $OUT
cd $R
( for i in 1 2 3 4 5 6 7 8 9 10 11 12; do sleep 4; cat /sys/class/drm/card*/device/pp_dpm_sclk 2>/dev/null | grep '\*' | head -2 | tr '\n' ' '; rocm-smi --showclocks 2>/dev/null | grep -i "sclk" | head -1; done ) > $OUT/clocks.txt 2>&1 &
CK=$!
timeout 400 python tools/pipe_batch_scaling.py 1.0 c2 $BS 2> $OUT/pipe_batch_scaling.txt > $OUT/pipe_batch_scaling.json; grep -v amdgpu.ids $OUT/pipe_batch_scaling.txt | cut -c1-300
for v in ${PIPE_AB:-}; do       # arms separated by blanks, the variables of one arm by commas; `base` = no variable
  echo "--- $v" >> $OUT/pipe_batch_scaling_ab.txt
  [ "$v" = base ] && e="" || e=$(echo $v | tr ',' ' ')
  env $e timeout 300 python tools/pipe_batch_scaling.py 1.0 c2 ${PIPE_AB_BS:-$BS} 2>> $OUT/pipe_batch_scaling_ab.txt > /dev/null
done
[ -f $OUT/pipe_batch_scaling_ab.txt ] && grep -v amdgpu.ids $OUT/pipe_batch_scaling_ab.txt | cut -c1-300
wait $CK; sort $OUT/clocks.txt | uniq -c | head -8
if [ -n "$3" ]; then
  python tacotron-2_amd/csrc/build.py --pipe-svc > /dev/null 2>&1
  timeout 300 python tools/pipe_svc_trace.py $3 > $OUT/pipe_svc_trace.txt 2>&1; grep -v amdgpu.ids $OUT/pipe_svc_trace.txt
fi
