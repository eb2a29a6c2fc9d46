#!/bin/bash
# rocprofv3 kernel stats of the bf16 bench step with the BatchNorm prologue on and off (same box, back to back).
# usage: tools/prologue_prof.sh   -> gpurun_out/prologue_prof/{on,off}_kernel_stats.csv
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
out=gpurun_out/prologue_prof; rm -rf $out; mkdir -p $out
for on in 1 0; do
  DODA_BN_PROLOGUE=$on timeout 400 rocprofv3 --kernel-trace --stats -f csv -d $out/p$on -o k -- python bench.py --no-cpu-baseline --fp32-steps 0 --kernel-reps 5 --steps 60 > $out/bench_$on.json 2> $out/p$on.err
  cp $out/p$on/k_kernel_stats.csv $out/prologue${on}_kernel_stats.csv 2>/dev/null
  rm -rf $out/p$on
done
python - <<PY
import csv
for on in (1, 0):
    rows = list(csv.DictReader(open("$out/prologue%d_kernel_stats.csv" % on)))
    steps = 90.0
    print("prologue=%d  total %.3f ms/step, %d launches/step" % (on, sum(float(r["TotalDurationNs"]) for r in rows) / steps / 1e6, sum(int(r["Calls"]) for r in rows) / steps))
    for r in rows:
        n = r["Name"]
        if "conv_tile" in n or "bn_apply" in n or "bn_fwd_final" in n or "bn_bwd" in n:
            print("   %6d calls %8.1f us/step  avg %6.1f us  %s" % (int(r["Calls"]), float(r["TotalDurationNs"]) / steps / 1e3, float(r["AverageNs"]) / 1e3, n[:90]))
PY
