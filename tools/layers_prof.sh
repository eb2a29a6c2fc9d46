#!/bin/bash
# rocprofv3 kernel statistics of the bench step per coarse backend (tools/layers_ab.py, one mode per run): gpurun_out/lay_<mode>_stats.csv
cd /tmp && export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-/root/repo}
for mode in ${MODES:-off layers}; do
  rm -rf /tmp/lp_$mode
  rocprofv3 --kernel-trace --stats -f csv -d /tmp/lp_$mode -o k -- python $R/tools/layers_ab.py --modes $mode --rounds 1 --steps 40 ${EXTRA} > $R/gpurun_out/lay_${mode}.log 2>&1
  f=$(find /tmp/lp_$mode -name "*kernel_stats.csv" | head -1)
  cp "$f" $R/gpurun_out/lay_${mode}_stats.csv
  grep "ms/step" $R/gpurun_out/lay_${mode}.log
  python $R/tools/kstats.py $R/gpurun_out/lay_${mode}_stats.csv 51 60 > $R/gpurun_out/lay_${mode}_kstats.txt
  head -14 $R/gpurun_out/lay_${mode}_kstats.txt
done
