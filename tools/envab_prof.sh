#!/bin/bash
# rocprofv3 kernel stats of the bf16 bench step with an environment switch on (1) and off (0), same box, back to back.
# usage: tools/envab_prof.sh DODA_SKIP_FUSION [pattern ...]   -> gpurun_out/envab_<VAR>/{1,0}_kernel_stats.csv + a summary
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
var=$1; shift; pat="${*:-add copy Cat bn_bwd}"
out=gpurun_out/envab_$var; rm -rf $out; mkdir -p $out
for on in ${ENVAB_VALUES:-1 0 1 0}; do
  env $var=$on timeout 400 rocprofv3 --kernel-trace --stats -f csv -d $out/p$on -o k -- python bench.py --no-cpu-baseline --fp32-steps 0 --kernel-reps 5 --steps 60 > $out/bench_$on.json 2> $out/p$on.err
  cp $out/p$on/k_kernel_stats.csv $out/${on}_kernel_stats.csv 2>/dev/null
  python - <<PY
import csv, json
rows = list(csv.DictReader(open("$out/${on}_kernel_stats.csv")))
steps = 90.0
side = ("subm_", "down2_", "pairs_", "scan_", "tilebook_build", "fillBuffer", "conv_assign")
tot = sum(float(r["TotalDurationNs"]) for r in rows if "spin_kernel" not in r["Name"])
main = sum(float(r["TotalDurationNs"]) for r in rows if "spin_kernel" not in r["Name"] and not any(s in r["Name"] for s in side))
ms = json.loads(open("$out/bench_$on.json").read().strip().splitlines()[-1])["ms_per_step"]
print("$var=$on  %.3f ms/step under rocprof; kernels %.3f ms/step (main stream ~%.3f), %.1f launches/step" % (ms, tot / steps / 1e6, main / steps / 1e6, sum(int(r["Calls"]) for r in rows) / steps))
for r in rows:
    if any(p in r["Name"] for p in "$pat".split()) and float(r["TotalDurationNs"]) / steps > 8000:
        print("   %5.1f/step %8.1f us/step  avg %6.1f us  %s" % (int(r["Calls"]) / steps, float(r["TotalDurationNs"]) / steps / 1e3, float(r["AverageNs"]) / 1e3, r["Name"][:100]))
PY
  rm -rf $out/p$on
done
