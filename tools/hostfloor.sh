# the bench step at sizes where the GPU has (nearly) nothing to do: what the issuing thread alone needs per step
cd /root/repo
mkdir -p gpurun_out/hf
run() { tag=$1; shift; env "$@" timeout 300 python bench.py --steps 100 --warmup 30 --fp32-steps 0 --no-train-entry --no-cpu-baseline --kernel-reps 2 $EXTRA > gpurun_out/hf/$tag.json 2> gpurun_out/hf/$tag.err
python - <<PY
import json
d=json.loads(open("gpurun_out/hf/$tag.json").read().strip().splitlines()[-1])
print("$tag: %.3f ms/step voxels/gpu %d" % (d["ms_per_step"], d["config"]["voxels_per_gpu"]))
PY
}
EXTRA="--voxels 5000" run small_default A=1
EXTRA="--voxels 5000" run small_exec DODA_COARSE_EXEC=1
EXTRA="--voxels 5000" run small_noprefetch A=1
EXTRA="--voxels 40000" run mid_default A=1
EXTRA="--voxels 40000" run mid_exec DODA_COARSE_EXEC=1
EXTRA="" run full_default A=1
EXTRA="" run full_exec DODA_COARSE_EXEC=1
EXTRA="" run full_exec4 DODA_COARSE_EXEC=1 DODA_COARSE_LEVEL=4
