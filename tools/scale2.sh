cd /root/repo
mkdir -p gpurun_out/scale
run() { tag=$1; shift; env "$@" timeout 300 python bench.py --steps 60 --warmup 20 --fp32-steps 0 --no-train-entry --no-cpu-baseline --kernel-reps 2 --config5-steps 0 $EXTRA > gpurun_out/scale/$tag.json 2> gpurun_out/scale/$tag.err
python - <<PY
import json
d=json.loads(open("gpurun_out/scale/$tag.json").read().strip().splitlines()[-1])
print("$tag: %.3f ms/step voxels %d" % (d["ms_per_step"], d["config"]["voxels_per_gpu"]))
PY
}
for r in 1 2; do
for s in 1 2 4 8 12; do
EXTRA="--scenes $s" run t_s${s}_$r A=1
done
done
