# the input layer's rows in one launch (DODA_INPUT_ROWS=1, default) against cat + pool + cast + pad (=0): alternating bench runs
cd /root/repo
mkdir -p gpurun_out/inrows
run() { tag=$1; shift; env "$@" timeout 300 python bench.py --steps 200 --warmup 30 --fp32-steps 0 --no-train-entry --no-cpu-baseline --kernel-reps 2 --config5-steps 0 $EXTRA > gpurun_out/inrows/$tag.json 2> gpurun_out/inrows/$tag.err
python - <<PY
import json
try:
    d=json.loads(open("gpurun_out/inrows/$tag.json").read().strip().splitlines()[-1])
    print("$tag: %.3f ms/step loss %.6f launches %s" % (d["ms_per_step"], d["config"]["final_loss"], d["config"].get("launches_per_step")))
except Exception as e:
    print("$tag: failed", e)
PY
}
for r in 1 2 3 4; do
EXTRA="" run four_$r DODA_INPUT_ROWS=0
EXTRA="" run one_$r DODA_INPUT_ROWS=1
done
EXTRA="--scenes 8" run four_s8 DODA_INPUT_ROWS=0
EXTRA="--scenes 8" run one_s8 DODA_INPUT_ROWS=1
EXTRA="--dtype f32 --steps 40" run four_f32 DODA_INPUT_ROWS=0
EXTRA="--dtype f32 --steps 40" run one_f32 DODA_INPUT_ROWS=1
