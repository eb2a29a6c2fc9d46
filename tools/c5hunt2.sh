# intermittent wrong final loss of the 1 cm B4 bench run seen on ONE box: reproduce, and if this box shows it, discriminate
cd /root/repo
mkdir -p gpurun_out/c5h
one() { tag=$1; shift; env "$@" timeout 300 python bench.py --voxel-scale 100 --voxels 500000 --steps 20 --warmup 8 --fp32-steps 0 --no-train-entry --no-cpu-baseline --kernel-reps 0 --config5-steps 0 $EXTRA > gpurun_out/c5h/$tag.json 2> gpurun_out/c5h/$tag.err
python - <<PY
import json
try:
    d=json.loads(open("gpurun_out/c5h/$tag.json").read().strip().splitlines()[-1])
    print("$tag %.6f %.2f" % (d["config"]["final_loss"], d["ms_per_step"]))
except Exception as e:
    print("$tag failed", open("gpurun_out/c5h/$tag.err").read()[-300:])
PY
}
uptime
rocm-smi --showclocks 2>/dev/null | grep -i -E "sclk|mclk" | head -4
bad=0
for r in 1 2 3 4 5 6 7 8 9 10; do
out=$(EXTRA="" one default_$r A=1); echo "$out"
case "$out" in *0.607937*) ;; *) bad=$((bad+1));; esac
done
echo "bad runs: $bad"
if [ $bad -gt 0 ]; then
for r in 1 2 3 4 5 6; do
EXTRA="--prefetch 0" one noprefetch_$r A=1
EXTRA="" one notile_$r DODA_NO_TILE=1
EXTRA="" one nowdma_$r DODA_NO_WDMA=1
EXTRA="" one nowlds_$r DODA_NO_WLDS=1
EXTRA="" one serial_$r AMD_SERIALIZE_KERNEL=3
done
fi
