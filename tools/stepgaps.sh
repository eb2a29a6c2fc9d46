#!/bin/bash
# Idle time of the main stream inside a training step: rocprofv3 kernel trace of a short bench; per step (between two sgd kernels)
# the main queue's busy time, the gaps between consecutive kernels on it, and the largest gaps with the kernels around them.
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
out=gpurun_out/gaps; rm -rf $out; mkdir -p $out
timeout 300 rocprofv3 --kernel-trace -f csv -d $out -o k -- python bench.py --no-cpu-baseline --fp32-steps 0 --kernel-reps 0 --steps 30 --warmup 10 > $out/bench.json 2> $out/err.txt
python - <<PY
import csv, collections
rows = list(csv.DictReader(open("$out/k_kernel_trace.csv")))
qs = collections.Counter(r["Queue_Id"] for r in rows)
main = qs.most_common(1)[0][0]
ev = sorted(((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"]) for r in rows if r["Queue_Id"] == main))
sgd = [k for k, e in enumerate(ev) if "sgd" in e[2]]
print("queues:", dict(qs), " main:", main, " steps seen:", len(sgd) - 1)
tot_busy = tot_gap = tot_len = 0.0; big = collections.Counter(); n = 0
for a, b in zip(sgd[10:-1], sgd[11:]):
    seg = ev[a:b + 1]
    busy = sum(e[1] - e[0] for e in seg[1:]) / 1e3
    length = (seg[-1][1] - seg[0][1]) / 1e3
    gaps = [(seg[k + 1][0] - seg[k][1]) / 1e3 for k in range(len(seg) - 1)]
    tot_busy += busy; tot_len += length; tot_gap += sum(g for g in gaps if g > 0); n += 1
    for k, g in enumerate(gaps):
        if g > 8.0: big[(seg[k][2][:48], seg[k + 1][2][:48])] += g
print("per step: %.1f us between sgd kernels, main queue busy %.1f us, gaps %.1f us (%d launches)" % (tot_len / n, tot_busy / n, tot_gap / n, len(ev[sgd[10]:sgd[11]])))
import statistics
allg = []
for a, b in zip(sgd[10:-1], sgd[11:]):
    seg = ev[a:b + 1]; allg += [(seg[k + 1][0] - seg[k][1]) / 1e3 for k in range(len(seg) - 1)]
allg = [g for g in allg if g > 0]
print("gap between consecutive kernels: median %.2f us, p90 %.2f, sum of gaps > 8 us per step: %.1f us" % (statistics.median(allg), sorted(allg)[int(0.9 * len(allg))], sum(g for g in allg if g > 8) / n))
for (x, y), g in big.most_common(8):
    print("   %7.1f us/step  after %-48s before %s" % (g / n, x, y))
PY
rm -f $out/*.csv
