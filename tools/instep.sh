# in-step averages of the statistics-producing kernels (rocprofv3 kernel stats of a short bench)
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
rm -rf /tmp/ins; timeout 400 rocprofv3 --kernel-trace --stats -f csv -d /tmp/ins -o k -- python bench.py --no-cpu-baseline --no-train-entry --config5-steps 0 --fp32-steps 0 --kernel-reps 2 --steps 60 --warmup 20 > /tmp/ins.json 2>/dev/null
python - <<PY
import csv, json
for r in csv.DictReader(open("/tmp/ins/k_kernel_stats.csv")):
    n = r["Name"]
    if any(k in n for k in ("conv_tile16<false, true>", "conv_tile<1, false, true, 2, true>", "conv_wlds48<true>", "PBF16W, 4, 2, 3, false, true, true", "wgrad_dma16(", "conv_fast<(anonymous namespace)::PBF16W, 1, 1, 3, false, true, false>")):
        print("%-70s %5s x %7.2f us" % (n.replace("(anonymous namespace)::", "")[:70], r["Calls"], float(r["AverageNs"]) / 1e3))
print("bench under rocprof: %.3f ms/step" % json.loads(open("/tmp/ins.json").read().strip().splitlines()[-1])["ms_per_step"])
PY
