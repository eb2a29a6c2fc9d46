# BatchNorm statistics as fp64 totals (one-launch BatchNorm, DODA_STATS_TOTALS=1, default) against rows + reduction launch (=0)
cd /root/repo
mkdir -p gpurun_out/tot
run() { tag=$1; shift; env "$@" timeout 300 python bench.py --steps 100 --warmup 30 --fp32-steps 0 --no-train-entry --no-cpu-baseline --kernel-reps 2 --config5-steps 0 $EXTRA > gpurun_out/tot/$tag.json 2> gpurun_out/tot/$tag.err
python - <<PY
import json
d=json.loads(open("gpurun_out/tot/$tag.json").read().strip().splitlines()[-1])
print("$tag: %.3f ms/step loss %.6f" % (d["ms_per_step"], d["config"]["final_loss"]))
PY
}
for r in 1 2 3; do
EXTRA="" run rows_$r DODA_STATS_TOTALS=0
EXTRA="" run totals_$r DODA_STATS_TOTALS=1
done
EXTRA="--scenes 8" run rows_s8 DODA_STATS_TOTALS=0
EXTRA="--scenes 8" run totals_s8 DODA_STATS_TOTALS=1
EXTRA="--scenes 1" run rows_s1 DODA_STATS_TOTALS=0
EXTRA="--scenes 1" run totals_s1 DODA_STATS_TOTALS=1
EXTRA="--dtype f32 --steps 40" run rows_f32 DODA_STATS_TOTALS=0
EXTRA="--dtype f32 --steps 40" run totals_f32 DODA_STATS_TOTALS=1
