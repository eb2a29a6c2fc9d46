# HIP runtime switches against the launch-bound step (432 launches in ~5 ms): kernel arguments in device memory, number of hardware
# queues, SDMA for the small copies.  Alternating runs of the default bench line.
cd /root/repo
mkdir -p gpurun_out/rtenv
run() { tag=$1; shift; env "$@" timeout 300 python bench.py --steps 100 --warmup 30 --fp32-steps 0 --no-train-entry --no-cpu-baseline --kernel-reps 2 --config5-steps 0 $EXTRA > gpurun_out/rtenv/$tag.json 2> gpurun_out/rtenv/$tag.err
python - <<PY
import json
try:
    d=json.loads(open("gpurun_out/rtenv/$tag.json").read().strip().splitlines()[-1])
    print("$tag: %.3f ms/step loss %.6f" % (d["ms_per_step"], d["config"]["final_loss"]))
except Exception as e:
    print("$tag: failed", e)
PY
}
for r in 1 2; do
EXTRA="" run base_$r A=1
EXTRA="" run kernarg1_$r HIP_FORCE_DEV_KERNARG=1
EXTRA="" run kernarg0_$r HIP_FORCE_DEV_KERNARG=0
EXTRA="" run queues2_$r GPU_MAX_HW_QUEUES=2
EXTRA="" run queues8_$r GPU_MAX_HW_QUEUES=8
EXTRA="" run nosdma_$r HSA_ENABLE_SDMA=0
done
EXTRA="--scenes 1" run base_s1 A=1
EXTRA="--scenes 1" run kernarg1_s1 HIP_FORCE_DEV_KERNARG=1
EXTRA="--scenes 1" run kernarg0_s1 HIP_FORCE_DEV_KERNARG=0
