#!/bin/bash
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
out=gpurun_out/bntrace; rm -rf $out; mkdir -p $out
timeout 300 rocprofv3 --kernel-trace -f csv -d $out -o k -- python tools/bnbench.py > $out/log.txt 2>&1
python - <<PY
import csv, collections
agg = collections.OrderedDict()
for r in csv.DictReader(open("$out/k_kernel_trace.csv")):
    n = r["Kernel_Name"].replace("(anonymous namespace)::", "").replace("void ", "")
    if "bn_" not in n: continue
    key = (n[:34], int(r["Grid_Size_X"]) // max(int(r["Workgroup_Size_X"]), 1))
    agg.setdefault(key, []).append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3)
for (n, g), v in agg.items():
    v = sorted(v); print("%-34s blocks %6d  n %3d  median %7.1f us" % (n, g, len(v), v[len(v) // 2]))
PY
rm -f $out/*.csv
