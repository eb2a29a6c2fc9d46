#!/bin/bash
# Round artifacts on the GPU box: default bench (bf16, with cpu_baseline), fp32 bench, rocprofv3 kernel
# stats of the default command, and FETCH_SIZE / WRITE_SIZE passes for the roofline kernel.
# usage: tools/profile_round.sh <tag>     -> gpurun_out/<tag>/
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
tag=${1:-r01}; out=gpurun_out/$tag; rm -rf $out; mkdir -p $out
timeout 600 python bench.py > $out/bench_bf16.json 2> $out/bench_bf16.err
timeout 300 python bench.py --dtype f32 --no-cpu-baseline > $out/bench_f32.json 2> $out/bench_f32.err
timeout 400 rocprofv3 --kernel-trace --stats -f csv -d $out/prof_bf16 -o k -- python bench.py --no-cpu-baseline > $out/bench_bf16_under_rocprof.json 2> $out/prof_bf16.err
timeout 400 rocprofv3 --kernel-trace --stats -f csv -d $out/prof_f32 -o k -- python bench.py --dtype f32 --no-cpu-baseline > $out/bench_f32_under_rocprof.json 2> $out/prof_f32.err
python tools/k1.py bf16 fwd 16 > /dev/null 2>&1   # warm the batch cache
for dt in bf16 f32; do
  for set in FETCH_SIZE WRITE_SIZE; do
    timeout 90 rocprofv3 --kernel-trace --pmc $set -f csv -d $out/pmc_${dt}_$set -o p -- python tools/k1.py $dt fwd 16 > $out/pmc_${dt}_$set.log 2>&1
  done
done
python - <<PY
import csv, glob, json, collections
res = {}
for dt in ("bf16", "f32"):
    d = {}
    for cs in ("FETCH_SIZE", "WRITE_SIZE"):
        vals = []
        for f in glob.glob("$out/pmc_%s_%s/*counter_collection.csv" % (dt, cs)):
            for r in csv.DictReader(open(f)):
                if "conv_fast" in r["Kernel_Name"] and r["Counter_Name"] == cs:
                    vals.append(float(r["Counter_Value"]))
        d[cs + "_KB_mean"] = sum(vals) / len(vals) if vals else None
        d[cs + "_n"] = len(vals)
    res[dt] = d
json.dump(res, open("$out/pmc_traffic_raw.json", "w"), indent=1)
print(json.dumps(res))
PY
for d in prof_bf16 prof_f32; do cp $out/$d/k_kernel_stats.csv $out/${d}_kernel_stats.csv 2>/dev/null; done
rm -rf $out/prof_bf16 $out/prof_f32 $out/pmc_*_SIZE
tail -c 600 $out/bench_bf16.json; echo; tail -c 300 $out/bench_f32.json
