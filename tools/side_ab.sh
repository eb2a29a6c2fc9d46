# A/B of the side flush of weight gradients (DODA_WGRAD_SIDE_LEVEL): bench ms/step per level, two rounds
cd /root/repo
mkdir -p gpurun_out/side
for round in 1 2; do
for lvl in 0 4 3 5 2; do
DODA_WGRAD_SIDE_LEVEL=$lvl timeout 300 python bench.py --steps 100 --warmup 30 --fp32-steps 0 --no-train-entry --no-cpu-baseline --kernel-reps 5 > gpurun_out/side/l${lvl}_r$round.json 2> gpurun_out/side/l${lvl}_r$round.err
python - <<PY
import json
d=json.loads(open("gpurun_out/side/l${lvl}_r$round.json").read().strip().splitlines()[-1])
print("level $lvl round $round: %.3f ms/step loss %.6f" % (d["ms_per_step"], d["config"]["final_loss"]))
PY
done
done
