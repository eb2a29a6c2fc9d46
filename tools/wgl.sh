#!/bin/bash
# per-instantiation times of wgrad_multi_kernel in the bench step (rocprofv3), us per launch
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
out=gpurun_out/wgl; rm -rf $out; mkdir -p $out
timeout 300 rocprofv3 --kernel-trace --stats -f csv -d $out -o k -- python bench.py --no-cpu-baseline --fp32-steps 0 --kernel-reps 5 --steps 60 --warmup 20 > $out/bench.json 2> $out/err.txt
grep "wgrad_multi_kernel\|wgrad_reduce_multi" $out/k_kernel_stats.csv | awk -F'",' '{split($2,a,","); n=$1; sub(/.*wgrad_/,"wgrad_",n); printf "  %-70s calls/step %.2f avg %.1f us\n", substr(n,1,70), a[1]/80, a[3]/1000; t+=a[2]/80000} END {printf "  total %.1f us per step\n", t}'
rm -rf $out/*.csv
