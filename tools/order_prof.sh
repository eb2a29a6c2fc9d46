# kernel time per step under both voxel numberings (8 scenes: GPU-bound), per kernel family
cd /tmp && export TMPDIR=/tmp
for o in first morton; do
rm -rf /tmp/prof_$o
timeout 600 rocprofv3 --kernel-trace --stats -d /tmp/prof_$o -o p --output-format csv -- python /root/repo/bench.py --voxel-order $o --scenes ${SCENES:-8} --steps 20 --warmup 10 --fp32-steps 0 --no-train-entry --no-cpu-baseline --kernel-reps 0 --config5-steps 0 > /dev/null 2>&1
mkdir -p /root/repo/gpurun_out/order
cp $(find /tmp/prof_$o -name "*kernel_stats.csv" | head -1) /root/repo/gpurun_out/order/stats_$o.csv
done
cd /root/repo
python - <<'PY'
import csv, re
def load(o):
    d = {}
    for r in csv.DictReader(open("gpurun_out/order/stats_%s.csv" % o)):
        d[r["Name"]] = (int(r["Calls"]), float(r["TotalDurationNs"]) / 1e3)
    return d
a, b = load("first"), load("morton")
rows = []
for k in set(a) | set(b):
    ca, ta = a.get(k, (0, 0.0)); cb, tb = b.get(k, (0, 0.0))
    rows.append((tb - ta, k, ca, ta, cb, tb))
rows.sort()
tot_a = sum(v[1] for v in a.values()); tot_b = sum(v[1] for v in b.values())
print("total kernel time: first %.0f us, morton %.0f us (30 steps)" % (tot_a, tot_b))
for d, k, ca, ta, cb, tb in rows[:12] + rows[-14:]:
    print("%+9.0f us  %-80s first %5d x %7.1f  morton %5d x %7.1f" % (d, k[:80], ca, ta / max(ca, 1), cb, tb / max(cb, 1)))
PY
