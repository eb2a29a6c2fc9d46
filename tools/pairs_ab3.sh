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
EXTRA="" run new_$r A=1
EXTRA="" run nopairs_$r DODA_WGRAD_PAIRS=0
EXTRA="" run down_$r DODA_WGRAD_PAIRS_DOWN=1
done
DODA_TRACE_PAIRS=1 timeout 300 python bench.py --steps 1 --warmup 1 --fp32-steps 0 --no-train-entry --no-cpu-baseline --kernel-reps 0 --config5-steps 0 2>&1 | grep -c "lazy pair"
