cd /root/repo
mkdir -p gpurun_out/hf
run() { tag=$1; shift; env "$@" timeout 300 python bench.py --steps 60 --warmup 20 --fp32-steps 0 --no-train-entry --no-cpu-baseline --kernel-reps 2 $EXTRA > gpurun_out/hf/$tag.json 2> gpurun_out/hf/$tag.err
python - <<PY
import json
try:
    d=json.loads(open("gpurun_out/hf/$tag.json").read().strip().splitlines()[-1])
    print("$tag: %.3f ms/step loss %.5f" % (d["ms_per_step"], d["config"]["final_loss"]))
except Exception as e:
    print("$tag: failed", open("gpurun_out/hf/$tag.err").read()[-400:])
PY
}
EXTRA="" run d A=1
EXTRA="" run e5_32x1 DODA_COARSE_EXEC=1 DODA_COARSE_LEVEL=5
EXTRA="" run e5_64x2 DODA_COARSE_EXEC=1 DODA_COARSE_LEVEL=5 DODA_CX_WGS=64 DODA_CX_XCDS=2
EXTRA="" run e5_128x4 DODA_COARSE_EXEC=1 DODA_COARSE_LEVEL=5 DODA_CX_WGS=128 DODA_CX_XCDS=4
EXTRA="" run e5_256x8 DODA_COARSE_EXEC=1 DODA_COARSE_LEVEL=5 DODA_CX_WGS=256 DODA_CX_XCDS=8
EXTRA="" run e6_128x4 DODA_COARSE_EXEC=1 DODA_COARSE_LEVEL=6 DODA_CX_WGS=128 DODA_CX_XCDS=4
EXTRA="" run e6_16x1 DODA_COARSE_EXEC=1 DODA_COARSE_LEVEL=6 DODA_CX_WGS=16 DODA_CX_XCDS=1
