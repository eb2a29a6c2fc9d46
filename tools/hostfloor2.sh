cd /root/repo
mkdir -p gpurun_out/hf
run() { tag=$1; shift; env "$@" timeout 300 python bench.py --steps 100 --warmup 30 --fp32-steps 0 --no-train-entry --no-cpu-baseline --kernel-reps 2 $EXTRA > gpurun_out/hf/$tag.json 2> gpurun_out/hf/$tag.err
python - <<PY
import json
d=json.loads(open("gpurun_out/hf/$tag.json").read().strip().splitlines()[-1])
print("$tag: %.3f ms/step voxels/gpu %d" % (d["ms_per_step"], d["config"]["voxels_per_gpu"]))
PY
}
for r in 1 2 3; do
EXTRA="" run full_default_$r A=1
EXTRA="" run full_exec6_$r DODA_COARSE_EXEC=1 DODA_COARSE_LEVEL=6
EXTRA="" run full_exec7_$r DODA_COARSE_EXEC=1 DODA_COARSE_LEVEL=7
EXTRA="" run full_exec6w64_$r DODA_COARSE_EXEC=1 DODA_COARSE_LEVEL=6 DODA_CX_WGS=64 DODA_CX_XCDS=2
done
