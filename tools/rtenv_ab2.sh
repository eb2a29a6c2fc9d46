# HIP_FORCE_DEV_KERNARG=1 against the default, alternating, at 1 scene (issue-bound) and 4 scenes (the bench)
cd /root/repo
mkdir -p gpurun_out/rtenv
run() { tag=$1; shift; env "$@" timeout 300 python bench.py --steps 200 --warmup 30 --fp32-steps 0 --no-train-entry --no-cpu-baseline --kernel-reps 2 --config5-steps 0 $EXTRA > gpurun_out/rtenv/$tag.json 2> gpurun_out/rtenv/$tag.err
python - <<PY
import json
try:
    d=json.loads(open("gpurun_out/rtenv/$tag.json").read().strip().splitlines()[-1])
    print("$tag: %.3f ms/step loss %.6f" % (d["ms_per_step"], d["config"]["final_loss"]))
except Exception as e:
    print("$tag: failed", e)
PY
}
for r in 1 2 3 4; do
EXTRA="--scenes 1" run b_s1_$r A=1
EXTRA="--scenes 1" run k1_s1_$r HIP_FORCE_DEV_KERNARG=1
EXTRA="" run b_s4_$r A=1
EXTRA="" run k1_s4_$r HIP_FORCE_DEV_KERNARG=1
done
