#!/bin/bash
# A/B of an environment switch on the weight-gradient kernels of the bench step (rocprofv3 kernel statistics, per step).
# usage: tools/wgab.sh ENVVAR v1 v2 ...
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
var=$1; shift
for v in "$@"; do
  out=gpurun_out/wgab_$v; rm -rf $out; mkdir -p $out
  env $var=$v timeout 300 rocprofv3 --kernel-trace --stats -f csv -d $out -o k -- python bench.py --no-cpu-baseline --fp32-steps 0 --kernel-reps 5 --steps 60 --warmup 20 > $out/bench.json 2> $out/err.txt
  python - <<PY
import csv, json
rows = list(csv.DictReader(open("$out/k_kernel_stats.csv")))
steps = 80.0
def tot(pat): return sum(float(r["TotalDurationNs"]) for r in rows if pat in r["Name"]) / 1e3 / steps
allk = sum(float(r["TotalDurationNs"]) for r in rows if "spin_kernel" not in r["Name"]) / 1e3 / steps
print("$var=$v: kernels %.0f us | wgrad_multi_kernel %.1f  wgrad_reduce_multi %.1f  wgrad_pairs %.1f  dma16 %.1f  dma_reduce %.1f  tilebook_build %.1f (us per step)" % (
    allk, tot("wgrad_multi_kernel"), tot("wgrad_reduce_multi"), tot("wgrad_pairs"), tot("wgrad_dma16"), tot("wgrad_dma_reduce"), tot("tilebook_build")))
PY
  rm -rf $out/*.csv
done
