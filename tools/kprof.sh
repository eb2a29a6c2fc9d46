#!/bin/bash
# usage: tools/kprof.sh <tag> [env assignments...] -- kbench args ; prints per-kernel avg durations
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
tag=$1; shift
envs=()
while [ "$1" != "--" ] && [ $# -gt 0 ]; do envs+=("$1"); shift; done
shift
out=gpurun_out/kprof_$tag; rm -rf $out; mkdir -p $out
env "${envs[@]}" timeout 300 rocprofv3 --kernel-trace --stats -f csv -d $out -o k -- python tools/kbench.py "$@" > $out/log.txt 2>&1
python - <<PY
import csv
rows=list(csv.DictReader(open("$out/k_kernel_stats.csv")))
for r in rows:
    n=r["Name"]
    if any(t in n for t in ("conv_", "wgrad", "pack_w", "subm_", "down2", "pairs", "vox", "scan")):
        short=n.replace("(anonymous namespace)::","").replace("void ","")[:70]
        print("%-70s calls %5s avg %9.1f us  min %9.1f" % (short, r["Calls"], float(r["AverageNs"])/1e3, float(r["MinNs"])/1e3))
PY
rm -f $out/*kernel_trace.csv $out/*agent_info.csv
