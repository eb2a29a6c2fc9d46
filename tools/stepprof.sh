#!/bin/bash
# usage: tools/stepprof.sh <tag> [env assignments...] : rocprofv3 kernel stats of a short bench run; prints per-step GPU time by kernel family
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
tag=$1; shift
out=gpurun_out/sp_$tag; rm -rf $out; mkdir -p $out
env "$@" timeout 300 rocprofv3 --kernel-trace --stats -f csv -d $out -o k -- python bench.py --no-cpu-baseline --fp32-steps 0 --kernel-reps 1 --steps 20 --warmup 5 > $out/bench.json 2> $out/err.txt
python - <<PY
import csv, json, collections
rows = list(csv.DictReader(open("$out/k_kernel_stats.csv")))
fam = collections.OrderedDict()
def family(n):
    for key in ("conv_fast", "conv_gather", "wgrad_multi", "wgrad_kernel", "wgrad_reduce", "bn_small", "bn_", "subm_", "down2", "pack_w", "voxel", "scan", "maxpool", "elementwise", "multi_tensor", "reduce", "fill", "copy", "softmax", "nll"):
        if key in n: return key
    return "other"
tot = 0
for r in rows:
    f = family(r["Name"]); d = fam.setdefault(f, [0, 0.0]); d[0] += int(r["Calls"]); d[1] += float(r["TotalDurationNs"]); tot += float(r["TotalDurationNs"])
steps = 25.0
print("$tag: GPU kernel time per step %.3f ms (25 steps incl. warmup)" % (tot / steps / 1e6))
for f, (c, t) in sorted(fam.items(), key=lambda kv: -kv[1][1]):
    print("   %-14s %6.1f launches/step %8.3f ms/step" % (f, c / steps, t / steps / 1e6))
try:
    b = json.loads(open("$out/bench.json").read().strip().splitlines()[-1]); print("   bench under rocprof: %.2f ms/step" % b["ms_per_step"])
except Exception as e: print("bench parse failed", e)
PY
rm -f $out/*kernel_trace.csv $out/*agent_info.csv
