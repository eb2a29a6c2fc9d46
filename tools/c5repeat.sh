cd /root/repo
mkdir -p gpurun_out/c5r
for r in 1 2 3 4 5 6; do
for t in 1 0; do
DODA_STATS_TOTALS=$t timeout 300 python bench.py --voxel-scale 100 --voxels 500000 --steps 20 --warmup 8 --fp32-steps 0 --no-train-entry --no-cpu-baseline --kernel-reps 2 --config5-steps 0 > gpurun_out/c5r/t${t}_$r.json 2> gpurun_out/c5r/t${t}_$r.err
python - <<PY
import json
d=json.loads(open("gpurun_out/c5r/t${t}_$r.json").read().strip().splitlines()[-1])
print("totals=$t run $r: %.3f ms/step loss %.6f" % (d["ms_per_step"], d["config"]["final_loss"]))
PY
done
done
