# step time against scenes per batch (default path and the coarse executor): the GPU-bound line and where the host takes over
cd /root/repo
mkdir -p gpurun_out/scale
run() { tag=$1; shift; env "$@" timeout 300 python bench.py --steps 50 --warmup 15 --fp32-steps 0 --no-train-entry --no-cpu-baseline --kernel-reps 2 $EXTRA > gpurun_out/scale/$tag.json 2> gpurun_out/scale/$tag.err
python - <<PY
import json
try:
    d=json.loads(open("gpurun_out/scale/$tag.json").read().strip().splitlines()[-1])
    print("$tag: %.3f ms/step voxels %d" % (d["ms_per_step"], d["config"]["voxels_per_gpu"]))
except Exception as e:
    print("$tag: failed", open("gpurun_out/scale/$tag.err").read()[-400:])
PY
}
for s in 1 2 3 4 6 8 12 16; do
EXTRA="--scenes $s" run d_s$s A=1
done
for s in 1 2 3 4; do
EXTRA="--scenes $s" run e6_s$s DODA_COARSE_EXEC=1 DODA_COARSE_LEVEL=6
EXTRA="--scenes $s" run e5_s$s DODA_COARSE_EXEC=1 DODA_COARSE_LEVEL=5
done
