# pair-list weight gradient of the strided rulebooks (DODA_WGRAD_PAIRS=1, default) against the gather-table kernel (=0): the
# export of the lists (pairs_count / pairs_scan / pairs_fill on the rulebook stream) against what the lists save
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
EXTRA="" run pairs_$r DODA_WGRAD_PAIRS=1
EXTRA="" run nopairs_$r DODA_WGRAD_PAIRS=0
done
EXTRA="--scenes 8" run pairs_s8 DODA_WGRAD_PAIRS=1
EXTRA="--scenes 8" run nopairs_s8 DODA_WGRAD_PAIRS=0
