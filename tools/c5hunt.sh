# intermittent wrong final loss of the 1 cm B4 bench run (Z-order numbering): which switch makes it go away?
cd /root/repo
mkdir -p gpurun_out/c5h
one() { tag=$1; shift; env "$@" timeout 300 python bench.py --voxel-scale 100 --voxels 500000 --steps 20 --warmup 8 --fp32-steps 0 --no-train-entry --no-cpu-baseline --kernel-reps 0 --config5-steps 0 $EXTRA > gpurun_out/c5h/$tag.json 2> gpurun_out/c5h/$tag.err
python - <<PY
import json
try:
    d=json.loads(open("gpurun_out/c5h/$tag.json").read().strip().splitlines()[-1])
    print("$tag %.6f" % d["config"]["final_loss"])
except Exception as e:
    print("$tag failed", open("gpurun_out/c5h/$tag.err").read()[-300:])
PY
}
for r in 1 2 3 4 5 6 7 8; do
EXTRA="" one default_$r A=1
EXTRA="--prefetch 0" one noprefetch_$r A=1
EXTRA="" one notile_$r DODA_NO_TILE=1
EXTRA="" one nowdma_$r DODA_NO_WDMA=1
done
