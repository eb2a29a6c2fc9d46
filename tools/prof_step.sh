#!/bin/bash
# usage: tools/prof_step.sh <tag> [bench args...] [-- ENV=..]: rocprofv3 --kernel-trace --stats of a short bench.py run (25 steps),
# per-step table by tools/kstats.py -> gpurun_out/ps_<tag>/{k_kernel_stats.csv,table.txt,bench.json}
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
tag=$1; shift
out=gpurun_out/ps_$tag; rm -rf $out; mkdir -p $out
timeout 400 rocprofv3 --kernel-trace --stats -f csv -d $out -o k -- python bench.py --no-cpu-baseline --fp32-steps 0 --kernel-reps 1 --steps 20 --warmup 5 "$@" > $out/bench.json 2> $out/err.txt
python tools/kstats.py $out/k_kernel_stats.csv 25 45 > $out/table.txt 2>&1
rm -f $out/*kernel_trace.csv $out/*agent_info.csv
cat $out/table.txt
