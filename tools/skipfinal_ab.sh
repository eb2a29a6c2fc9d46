cd /root/repo
mkdir -p gpurun_out/sf
run() { tag=$1; shift; env "$@" timeout 300 python bench.py --steps 100 --warmup 30 --fp32-steps 0 --no-train-entry --no-cpu-baseline --kernel-reps 2 --config5-steps 0 $EXTRA > gpurun_out/sf/$tag.json 2> gpurun_out/sf/$tag.err
python - <<PY
import json
d=json.loads(open("gpurun_out/sf/$tag.json").read().strip().splitlines()[-1])
print("$tag: %.3f ms/step" % d["ms_per_step"])
PY
}
for r in 1 2 3; do
EXTRA="" run base_$r A=1
EXTRA="" run skip_$r DODA_DEBUG_SKIP_FINAL=1
done
EXTRA="--scenes 8" run base_s8 A=1
EXTRA="--scenes 8" run skip_s8 DODA_DEBUG_SKIP_FINAL=1
EXTRA="--scenes 1" run base_s1 A=1
EXTRA="--scenes 1" run skip_s1 DODA_DEBUG_SKIP_FINAL=1
