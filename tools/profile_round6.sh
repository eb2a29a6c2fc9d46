#!/bin/bash
# Round-6 artifacts on the GPU box: default bench line, rocprofv3 kernel stats of the same step (bf16 + fp32), and
# FETCH_SIZE / WRITE_SIZE passes (separate --pmc runs, kernel-trace only) for the roofline kernel in the form the step
# launches it (conv_tile<0,false,true>: statistics + residual) and for the tile weight gradient (wgrad_dma16).
# usage: tools/profile_round6.sh [tag]   -> gpurun_out/<tag>/
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
tag=${1:-r06}; out=gpurun_out/$tag; rm -rf $out; mkdir -p $out
timeout 400 rocprofv3 --kernel-trace --stats -f csv -d $out/prof_bf16 -o k -- python bench.py --no-cpu-baseline --no-train-entry --config5-steps 0 --fp32-steps 0 --refgraph-steps 0 --kernel-reps 0 > $out/bench_bf16_under_rocprof.json 2> $out/prof_bf16.err
timeout 400 rocprofv3 --kernel-trace --stats -f csv -d $out/prof_f32 -o k -- python bench.py --dtype f32 --no-cpu-baseline --no-train-entry --config5-steps 0 --refgraph-steps 0 --steps 40 --warmup 10 --kernel-reps 0 > $out/bench_f32_under_rocprof.json 2> $out/prof_f32.err
python tools/k1.py bf16 fwd 16 > /dev/null 2>&1   # warm the batch cache
timeout 200 rocprofv3 --kernel-trace --stats -f csv -d $out/prof_k1 -o k -- python tools/k1.py bf16 fwdstep 16 50 > /dev/null 2> $out/prof_k1.err
for which in fwdstep wgradt; do
  for set in FETCH_SIZE WRITE_SIZE; do
    timeout 120 rocprofv3 --kernel-trace --pmc $set -f csv -d $out/pmc_${which}_$set -o p -- python tools/k1.py bf16 $which 16 > $out/pmc_${which}_$set.log 2>&1
  done
done
python - <<PY
import csv, glob, json
res = {}
def mean(pattern, kernel, cs):
    vals = []
    for f in glob.glob(pattern):
        for r in csv.DictReader(open(f)):
            if any(k in r["Kernel_Name"] for k in kernel.split("|")) and r["Counter_Name"] == cs:
                vals.append(float(r["Counter_Value"]))
    return (sum(vals) / len(vals) if vals else None), len(vals)
for which, kern, key in (("fwdstep", "conv_tile", "bf16"), ("wgradt", "wgrad_dma16", "wgrad_dma16_bf16_8_layers_per_launch")):
    d = {}
    for cs in ("FETCH_SIZE", "WRITE_SIZE"):
        d[cs + "_KB_mean"], d[cs + "_n"] = mean("$out/pmc_%s_%s/*counter_collection.csv" % (which, cs), kern, cs)
    d["kernel"] = kern
    res[key] = d
json.dump(res, open("$out/pmc_traffic_raw.json", "w"), indent=1)
print(json.dumps(res))
PY
for d in prof_bf16 prof_f32 prof_k1; do cp $out/$d/k_kernel_stats.csv $out/${d}_kernel_stats.csv 2>/dev/null; done
rm -rf $out/prof_bf16 $out/prof_f32 $out/prof_k1 $out/pmc_*_SIZE
# the default bench line quotes this round's in-step average and PMC traffic: put them where bench.py looks, then run it
mkdir -p profiles
cp $out/prof_bf16_kernel_stats.csv profiles/${tag}_bf16_kernel_stats.csv
cp $out/prof_f32_kernel_stats.csv profiles/${tag}_f32_kernel_stats.csv
cp $out/prof_k1_kernel_stats.csv profiles/${tag}_k1_fwdstep_kernel_stats.csv
cp $out/pmc_traffic_raw.json profiles/${tag}_pmc_traffic_raw.json
cp $out/bench_bf16_under_rocprof.json profiles/${tag}_bench_bf16_under_rocprof.json
timeout 1200 python bench.py > $out/bench_default.json 2> $out/bench_default.err
tail -c 300 $out/bench_default.json; echo
