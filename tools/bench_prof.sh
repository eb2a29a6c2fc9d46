#!/bin/bash
# rocprofv3 kernel statistics of a short default bench step: gpurun_out/bench_kstats.txt (tools/kstats.py)
cd /tmp && export TMPDIR=/tmp && cd ${GRAFT_REPO_ROOT:-/root/repo}
rm -rf /tmp/bp; STEPS=${STEPS:-40}; WARM=${WARM:-10}
rocprofv3 --kernel-trace --stats -f csv -d /tmp/bp -o k -- python bench.py --no-cpu-baseline --no-train-entry --config5-steps 0 --fp32-steps 0 --refgraph-steps 0 --kernel-reps 0 --steps $STEPS --warmup $WARM "$@" > gpurun_out/bench_prof.json 2> gpurun_out/bench_prof.err
f=$(find /tmp/bp -name "*kernel_stats.csv" | head -1)
cp "$f" gpurun_out/bench_kernel_stats.csv
python tools/kstats.py gpurun_out/bench_kernel_stats.csv $((STEPS + WARM + 1)) 70 > gpurun_out/bench_kstats.txt
head -16 gpurun_out/bench_kstats.txt
