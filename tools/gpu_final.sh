#!/bin/bash
# full evidence session + the other workloads of DESIGN 3.5
TAG=${1:-final}
R=${GRAFT_REPO_ROOT:-$(pwd)}
bash $R/tools/gpu_r2.sh $TAG
OUT=$R/gpurun_out/$TAG
cd $R
for w in default_hparams c2_4stack c5_stress; do
  timeout 300 python bench.py --workload $w --steps 20 --warmup 5 --no-cpu-baseline --no-synth --sustained 0 > $OUT/bench_$w.json 2> $OUT/bench_$w.err
  python -c "import json; d=json.loads(open('$OUT/bench_$w.json').read()); print('$w', '%.3f ms/step' % d['ms_per_step'], '%.3g samples/s' % d['value'])"
done
