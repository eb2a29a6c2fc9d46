cd /root/repo
mkdir -p gpurun_out/pa
run() { tag=$1; shift; env "$@" timeout 300 python bench.py --steps 100 --warmup 30 --fp32-steps 0 --no-train-entry --no-cpu-baseline --kernel-reps 2 --config5-steps 0 $EXTRA > gpurun_out/pa/$tag.json 2> gpurun_out/pa/$tag.err
python - <<PY
import json
d=json.loads(open("gpurun_out/pa/$tag.json").read().strip().splitlines()[-1])
print("$tag: %.3f ms/step loss %.6f" % (d["ms_per_step"], d["config"]["final_loss"]))
PY
}
for r in 1 2 3; do
EXTRA="" run down1_$r DODA_WGRAD_PAIRS_DOWN=1
EXTRA="" run down0_$r DODA_WGRAD_PAIRS_DOWN=0
done
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/pp; timeout 300 rocprofv3 --kernel-trace --stats -f csv -d /tmp/pp -o k -- python /root/repo/bench.py --steps 20 --warmup 10 --fp32-steps 0 --no-train-entry --no-cpu-baseline --kernel-reps 0 --config5-steps 0 > /dev/null 2>&1
grep -E "pairs_|wgrad_pairs|wgrad_multi" /tmp/pp/k_kernel_stats.csv | cut -d, -f1-4 | cut -c1-120
