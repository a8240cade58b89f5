R=${GRAFT_REPO_ROOT:-$(pwd)}; OUT=$R/gpurun_out/r6d; mkdir -p $OUT; cd $R
BQ="--no-cpu-baseline --no-synth --no-exclusive --no-other-workloads --sustained 0 --no-feeder"
for rep in 1 2; do
for w in c5_stress c2_4stack; do
  for g in 1 0; do
    WN_GEMM8P=$g timeout 300 python bench.py --workload $w --steps 20 --warmup 5 $BQ 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$w gemm8p=$g', '%.3f ms/step' % d['ms_per_step'])" >> $OUT/ab_workloads.txt 2>&1
  done
done
done
cat $OUT/ab_workloads.txt
