# direct-address rulebook grid up to 2^28 cells (default) against 2^26 (1 cm batches on the hash): config 5 B4 / B1, and the 2 cm bench
cd /root/repo
mkdir -p gpurun_out/grid
run() { tag=$1; shift; env "$@" timeout 300 python bench.py --fp32-steps 0 --no-train-entry --no-cpu-baseline --kernel-reps 0 --config5-steps 0 $EXTRA > gpurun_out/grid/$tag.json 2> gpurun_out/grid/$tag.err
python - <<PY
import json
try:
    d=json.loads(open("gpurun_out/grid/$tag.json").read().strip().splitlines()[-1])
    print("$tag: %.3f ms/step loss %.6f" % (d["ms_per_step"], d["config"]["final_loss"]))
except Exception as e:
    print("$tag failed", open("gpurun_out/grid/$tag.err").read()[-400:])
PY
}
for r in 1 2; do
EXTRA="--voxel-scale 100 --voxels 500000 --steps 20 --warmup 8" run c5b4_26_$r DODA_RULEBOOK_GRID_MAX_LOG2=26
EXTRA="--voxel-scale 100 --voxels 500000 --steps 20 --warmup 8" run c5b4_28_$r DODA_RULEBOOK_GRID_MAX_LOG2=28
EXTRA="--voxel-scale 100 --voxels 500000 --scenes 1 --steps 30 --warmup 10" run c5b1_26_$r DODA_RULEBOOK_GRID_MAX_LOG2=26
EXTRA="--voxel-scale 100 --voxels 500000 --scenes 1 --steps 30 --warmup 10" run c5b1_28_$r DODA_RULEBOOK_GRID_MAX_LOG2=28
done
EXTRA="--scenes 16 --steps 30 --warmup 10" run s16_26 DODA_RULEBOOK_GRID_MAX_LOG2=26
EXTRA="--scenes 16 --steps 30 --warmup 10" run s16_28 DODA_RULEBOOK_GRID_MAX_LOG2=28
