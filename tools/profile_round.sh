#!/bin/bash
# Round artifacts on the GPU box: the default bench (bf16 headline + fp32 sub-record + cpu_baseline),
# rocprofv3 kernel stats of the same training step, and FETCH_SIZE / WRITE_SIZE passes for the roofline
# kernel (level-1 SubM 16->16 gather) and for the pair-list weight gradient.
# usage: tools/profile_round.sh <tag>     -> gpurun_out/<tag>/
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
tag=${1:-r02}; out=gpurun_out/$tag; rm -rf $out; mkdir -p $out
timeout 900 python bench.py > $out/bench_default.json 2> $out/bench_default.err
timeout 400 rocprofv3 --kernel-trace --stats -f csv -d $out/prof_bf16 -o k -- python bench.py --no-cpu-baseline --fp32-steps 0 > $out/bench_bf16_under_rocprof.json 2> $out/prof_bf16.err
timeout 400 rocprofv3 --kernel-trace --stats -f csv -d $out/prof_f32 -o k -- python bench.py --dtype f32 --no-cpu-baseline --steps 40 --warmup 10 > $out/bench_f32_under_rocprof.json 2> $out/prof_f32.err
python tools/k1.py bf16 fwd 16 > /dev/null 2>&1   # warm the batch cache
for dt in bf16 f32; do
  for set in FETCH_SIZE WRITE_SIZE; do
    timeout 90 rocprofv3 --kernel-trace --pmc $set -f csv -d $out/pmc_${dt}_$set -o p -- python tools/k1.py $dt fwd 16 > $out/pmc_${dt}_$set.log 2>&1
  done
done
for set in FETCH_SIZE WRITE_SIZE; do
  timeout 90 rocprofv3 --kernel-trace --pmc $set -f csv -d $out/pmc_wgradp_$set -o p -- python tools/k1.py bf16 wgradp 16 > $out/pmc_wgradp_$set.log 2>&1
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
for dt in ("bf16", "f32"):
    d = {}
    for cs in ("FETCH_SIZE", "WRITE_SIZE"):
        d[cs + "_KB_mean"], d[cs + "_n"] = mean("$out/pmc_%s_%s/*counter_collection.csv" % (dt, cs), "conv_tile|conv_fast", cs)
    res[dt] = d
d = {}
for cs in ("FETCH_SIZE", "WRITE_SIZE"):
    d[cs + "_KB_mean"], d[cs + "_n"] = mean("$out/pmc_wgradp_%s/*counter_collection.csv" % cs, "wgrad_pairs_kernel", cs)
res["wgrad_pairs_bf16_8_layers_per_launch"] = d
json.dump(res, open("$out/pmc_traffic_raw.json", "w"), indent=1)
print(json.dumps(res))
PY
for d in prof_bf16 prof_f32; do cp $out/$d/k_kernel_stats.csv $out/${d}_kernel_stats.csv 2>/dev/null; done
rm -rf $out/prof_bf16 $out/prof_f32 $out/pmc_*_SIZE
tail -c 400 $out/bench_default.json; echo
