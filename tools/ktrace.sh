#!/bin/bash
# usage: tools/ktrace.sh <tag> [env...] -- <kbench args> : per-(kernel, grid) mean durations from a kernel trace
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
tag=$1; shift
envs=()
while [ "$1" != "--" ] && [ $# -gt 0 ]; do envs+=("$1"); shift; done
shift
out=gpurun_out/kt_$tag; rm -rf $out; mkdir -p $out
env "${envs[@]}" timeout 300 rocprofv3 --kernel-trace -f csv -d $out -o k -- python tools/kbench.py "$@" > $out/log.txt 2>&1
python - <<PY
import csv, collections
agg = collections.OrderedDict()
for r in csv.DictReader(open("$out/k_kernel_trace.csv")):
    n = r["Kernel_Name"].replace("(anonymous namespace)::", "").replace("void ", "")
    if not any(t in n for t in ("conv_fast", "wgrad", "bn_")): continue
    key = (n[:58], int(r["Grid_Size_X"]) // max(int(r["Workgroup_Size_X"]), 1))
    agg.setdefault(key, []).append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3)
for (n, g), v in agg.items():
    v = sorted(v)
    print("%-58s blocks %6d  n %4d  median %8.1f us  min %8.1f" % (n, g, len(v), v[len(v) // 2], v[0]))
PY
rm -f $out/*.csv
