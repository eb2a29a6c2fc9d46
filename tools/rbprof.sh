#!/bin/bash
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
out=gpurun_out/rbprof; rm -rf $out; mkdir -p $out
timeout 200 rocprofv3 --kernel-trace --memory-copy-trace --stats -f csv -d $out -o k -- python tools/rbtime.py > $out/log.txt 2>&1
python - <<PY
import csv
rows=list(csv.DictReader(open("$out/k_kernel_stats.csv")))
for r in rows[:16]:
    print("%-80s calls %5s avg %8.1f us total %8.1f us" % (r["Name"].replace("(anonymous namespace)::","")[:80], r["Calls"], float(r["AverageNs"])/1e3, float(r["TotalDurationNs"])/1e3))
PY
tail -3 $out/log.txt
rm -f $out/*trace.csv $out/*agent_info.csv
