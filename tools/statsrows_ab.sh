# DODA_STATS_MIN_ROWS: from how many output rows on a conv epilogue carries the BatchNorm statistics (below: bn_small_* one-launch kernels)
cd /root/repo
mkdir -p gpurun_out/sr
run() { tag=$1; shift; env "$@" timeout 300 python bench.py --steps 100 --warmup 30 --fp32-steps 0 --no-train-entry --no-cpu-baseline --kernel-reps 2 --config5-steps 0 $EXTRA > gpurun_out/sr/$tag.json 2> gpurun_out/sr/$tag.err
python - <<PY
import json
try:
    d=json.loads(open("gpurun_out/sr/$tag.json").read().strip().splitlines()[-1])
    print("$tag: %.3f ms/step loss %.6f" % (d["ms_per_step"], d["config"]["final_loss"]))
except Exception as e:
    print("$tag failed", open("gpurun_out/sr/$tag.err").read()[-500:])
PY
}
for r in 1 2 3; do
for v in 4096 1024 256 0; do
EXTRA="" run min${v}_$r DODA_STATS_MIN_ROWS=$v
done
done
