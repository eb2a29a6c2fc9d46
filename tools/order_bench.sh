cd /root/repo
mkdir -p gpurun_out/order
run() { tag=$1; shift; timeout 400 python bench.py --steps 60 --warmup 20 --fp32-steps 0 --no-train-entry --no-cpu-baseline --kernel-reps 30 "$@" > gpurun_out/order/$tag.json 2> gpurun_out/order/$tag.err
python - <<PY
import json
try:
    d=json.loads(open("gpurun_out/order/$tag.json").read().strip().splitlines()[-1])
    r=d["roofline"]
    print("$tag: %.3f ms/step  roofline %s frac %.3f frac_8d %s achieved %.0f in-step %s" % (d["ms_per_step"], r.get("kernel","")[:24], r["frac"], r.get("frac_8d"), r["achieved"], r.get("in_step_avg_us")))
    print("    gates:", {k: v for k, v in r.items() if "gate" in k})
except Exception as e:
    print("$tag: failed", e, open("gpurun_out/order/$tag.err").read()[-600:])
PY
}
for r in 1 2; do
run first_s4_$r --voxel-order first
run morton_s4_$r --voxel-order morton
run first_s8_$r --voxel-order first --scenes 8
run morton_s8_$r --voxel-order morton --scenes 8
done
