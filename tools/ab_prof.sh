#!/bin/bash
# kernel-stat A/B of one environment switch: tools/ab_prof.sh VAR  -> gpurun_out/ab_<VAR>_{0,1}_kernel_stats.csv
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
var=$1
for v in 0 1; do
  out=gpurun_out/ab_${var}_$v; rm -rf $out; mkdir -p $out
  env $var=$v timeout 400 rocprofv3 --kernel-trace --stats -f csv -d $out/p -o k -- python bench.py --no-cpu-baseline --fp32-steps 0 --kernel-reps 1 --steps 60 --warmup 20 > $out/bench.json 2> $out/err.txt
  cp $out/p/k_kernel_stats.csv gpurun_out/ab_${var}_${v}_kernel_stats.csv; rm -rf $out/p
done
