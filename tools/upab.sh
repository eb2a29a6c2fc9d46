#!/bin/bash
# conv_up32 against conv_fast in the bench step (rocprofv3): kernel time per step of the conv_up32 / K = 8 conv_fast rows
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
for v in 0 1; do
out=gpurun_out/upab_$v; rm -rf $out; mkdir -p $out
DODA_CONV_UP=$v timeout 300 rocprofv3 --kernel-trace --stats -f csv -d $out -o k -- python bench.py --no-cpu-baseline --fp32-steps 0 --kernel-reps 5 --steps 60 --warmup 20 > $out/bench.json 2> $out/err.txt
python - <<PY
import csv
rows = list(csv.DictReader(open("$out/k_kernel_stats.csv")))
steps = 80.0
def tot(pat): return sum(float(r["TotalDurationNs"]) for r in rows if pat in r["Name"]) / 1e3 / steps
allk = sum(float(r["TotalDurationNs"]) for r in rows if "spin_kernel" not in r["Name"]) / 1e3 / steps
print("DODA_CONV_UP=$v: kernels %.0f us | conv_fast %.1f  conv_up32 %.1f (us per step)" % (allk, tot("conv_fast"), tot("conv_up32")))
for r in rows:
    if "conv_up32" in r["Name"]:
        print("    %s  %.2f x %.1f us" % (r["Name"][24:60], float(r["Calls"]) / steps, float(r["AverageNs"]) / 1e3))
PY
rm -rf $out/*.csv
done
