cd /root/repo
mkdir -p gpurun_out/order
run() { tag=$1; shift; timeout 400 python bench.py --steps 100 --warmup 30 --fp32-steps 0 --no-train-entry --no-cpu-baseline --kernel-reps 2 --config5-steps 0 "$@" > gpurun_out/order/$tag.json 2> gpurun_out/order/$tag.err
python - <<PY
import json
d=json.loads(open("gpurun_out/order/$tag.json").read().strip().splitlines()[-1])
print("$tag: %.3f ms/step" % d["ms_per_step"])
PY
}
for r in 1 2 3; do
run first_s4_$r --voxel-order first
run morton_s4_$r --voxel-order morton
done
for r in 1 2; do
run first_s8_$r --voxel-order first --scenes 8
run morton_s8_$r --voxel-order morton --scenes 8
done
