#!/bin/bash
# Round 6, session a: the new parity tests at the benched synthesis geometries, smoke, hwmon probe, launch-class ablations of the live step
# (diagnostic build), then the default-flag bench line.
R=${GRAFT_REPO_ROOT:-$(pwd)}; cd $R
TAG=r9a; OUT=$R/gpurun_out/$TAG; mkdir -p $OUT
bash tools/gpu_session.sh $TAG "tests:test_hip_round6" smoke
( for d in /sys/class/drm/card*/device/hwmon/hwmon*; do echo "== $d"; ls $d; for f in power1_average power1_input energy1_input freq1_input freq2_input power1_cap; do [ -r $d/$f ] && echo "$f = $(cat $d/$f)"; done; done ) > $OUT/hwmon_probe.txt 2>&1
head -40 $OUT/hwmon_probe.txt
bash tools/gpu_session.sh $TAG ab:base=
python tacotron-2_amd/csrc/build.py --ablate > $OUT/build_ablate.log 2>&1; tail -1 $OUT/build_ablate.log
for m in 0 1 2 3 4 8 12 15; do bash tools/gpu_session.sh $TAG "ab:ablate$m=WN_ABLATE=$m"; done
rm -f tacotron-2_amd/csrc/wn_train.o; python tacotron-2_amd/csrc/build.py > $OUT/build_restore.log 2>&1; tail -1 $OUT/build_restore.log
bash tools/gpu_session.sh $TAG ab:base2=
bash tools/gpu_session.sh $TAG bench
python - <<'PY'
import json,os
d=json.load(open(os.path.join(os.environ.get('GRAFT_REPO_ROOT','.'),'gpurun_out/r9a/bench.json')))
print(json.dumps(d.get('synthesis_summary')))
print(json.dumps(d.get('synthesis_parity_gate')))
r=d['roofline']; print({k:r.get(k) for k in ('achieved','frac','sclk_in_kernel_mhz','peak_at_clock','frac_of_peak_at_clock')})
print(d['ms_per_step'], d.get('sustained',{}).get('ms_per_step'), d.get('sustained',{}).get('joules_per_step'))
PY
