cd /root/repo
mkdir -p gpurun_out/grid
run() { tag=$1; shift; env "$@" timeout 300 python bench.py --fp32-steps 0 --no-train-entry --no-cpu-baseline --kernel-reps 0 --config5-steps 0 --voxel-scale 100 --voxels 500000 --steps 20 --warmup 8 > gpurun_out/grid/$tag.json 2> gpurun_out/grid/$tag.err
python - <<PY
import json
d=json.loads(open("gpurun_out/grid/$tag.json").read().strip().splitlines()[-1])
print("$tag: %.3f ms/step loss %.6f" % (d["ms_per_step"], d["config"]["final_loss"]))
PY
}
for r in 1 2; do
run c5_pairs1_$r DODA_WGRAD_PAIRS=1
run c5_pairs0_$r DODA_WGRAD_PAIRS=0
done
