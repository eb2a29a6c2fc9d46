cd /root/repo
mkdir -p gpurun_out/side
run() { tag=$1; shift; env "$@" timeout 300 python bench.py --steps 60 --warmup 20 --fp32-steps 0 --no-train-entry --no-cpu-baseline --kernel-reps 2 $EXTRA > gpurun_out/side/$tag.json 2> gpurun_out/side/$tag.err
python - <<PY
import json
d=json.loads(open("gpurun_out/side/$tag.json").read().strip().splitlines()[-1])
print("$tag: %.3f ms/step voxels/gpu %d" % (d["ms_per_step"], d["config"]["voxels_per_gpu"]))
PY
}
for r in 1 2; do
EXTRA="--scenes 8" run s8_side0_$r DODA_WGRAD_SIDE_LEVEL=0
EXTRA="--scenes 8" run s8_side4_$r DODA_WGRAD_SIDE_LEVEL=4
EXTRA="--scenes 8" run s8_side3_$r DODA_WGRAD_SIDE_LEVEL=3
done
