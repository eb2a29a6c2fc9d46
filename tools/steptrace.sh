#!/bin/bash
# usage: tools/steptrace.sh <tag> [env...] : kernel trace of a short bf16 bench; per (kernel, blocks) time per step, top 45
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
tag=$1; shift
out=gpurun_out/st_$tag; rm -rf $out; mkdir -p $out
env "$@" timeout 300 rocprofv3 --kernel-trace -f csv -d $out -o k -- python bench.py --no-cpu-baseline --fp32-steps 0 --kernel-reps 1 --steps 20 --warmup 5 > $out/bench.json 2> $out/err.txt
python - <<PY
import csv, collections, re
agg = collections.OrderedDict()
tot = 0.0
for r in csv.DictReader(open("$out/k_kernel_trace.csv")):
    n = r["Kernel_Name"].replace("(anonymous namespace)::", "").replace("void ", "")
    n = re.sub(r"at::native::", "", n)
    key = (n[:70], int(r["Grid_Size_X"]) * int(r.get("Grid_Size_Y", 1) or 1) // max(int(r["Workgroup_Size_X"]), 1))
    d = (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3
    a = agg.setdefault(key, [0, 0.0]); a[0] += 1; a[1] += d; tot += d
steps = 25.0
print("$tag: %.3f ms GPU kernel time per step, %.0f launches per step" % (tot / steps / 1e3, sum(a[0] for a in agg.values()) / steps))
for (n, g), (c, t) in sorted(agg.items(), key=lambda kv: -kv[1][1])[:45]:
    print("%-70s blocks %6d  %5.1f/step  avg %7.1f us  %7.1f us/step" % (n, g, c / steps, t / c, t / steps))
PY
rm -f $out/*.csv
