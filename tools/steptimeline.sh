#!/bin/bash
# One bench step on the main queue as a timeline: per kernel its duration and the idle gap in front of it; totals per kernel family.
# usage: tools/steptimeline.sh [extra bench.py flags]   -> gpurun_out/timeline.txt
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
out=gpurun_out/tl; rm -rf $out; mkdir -p $out
timeout 300 rocprofv3 --kernel-trace -f csv -d $out -o k -- python bench.py --no-cpu-baseline --no-train-entry --config5-steps 0 --fp32-steps 0 --kernel-reps 0 --steps 12 --warmup 8 "$@" > $out/bench.json 2> $out/err.txt
python - > gpurun_out/timeline.txt <<PY
import csv, collections, re
rows = list(csv.DictReader(open("$out/k_kernel_trace.csv")))
qs = collections.Counter(r["Queue_Id"] for r in rows)
main = qs.most_common(1)[0][0]
ev = sorted(((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"], r["Queue_Id"]) for r in rows))
mainev = [e for e in ev if e[3] == main]
sgd = [k for k, e in enumerate(mainev) if "sgd" in e[2]]
a, b = sgd[-3], sgd[-2]
def short(n):
    n = re.sub(r"\(anonymous namespace\)::", "", n); n = re.sub(r"^void ", "", n)
    return n.split("(")[0][:64]
step = mainev[a + 1:b + 1]
span = (step[-1][1] - mainev[a][1]) / 1e3
busy = sum(e[1] - e[0] for e in step) / 1e3
gaps = [(step[k][0] - (step[k - 1][1] if k else mainev[a][1])) / 1e3 for k in range(len(step))]
print("main-queue step: %d kernels, span %.1f us, busy %.1f us, idle %.1f us (mean gap %.2f us)" % (len(step), span, busy, span - busy, (span - busy) / len(step)))
fam = collections.defaultdict(lambda: [0, 0.0, 0.0])
for e, g in zip(step, gaps):
    f = fam[short(e[2])]; f[0] += 1; f[1] += (e[1] - e[0]) / 1e3; f[2] += max(g, 0.0)
print("per kernel: count, busy us, idle us in front")
for k, v in sorted(fam.items(), key=lambda kv: -(kv[1][1] + kv[1][2]))[:45]:
    print("  %3d x  busy %7.1f  idle %6.1f  %s" % (v[0], v[1], v[2], k))
t0, t1 = mainev[a][1], step[-1][1]
print("glue launches inside the step window, per queue (main = %s):" % main)
gl = collections.Counter((e[3], short(e[2])) for e in ev if t0 <= e[0] <= t1 and re.search("fillBuffer|copyBuffer|at::native", e[2]))
for (q, n), c in sorted(gl.items()): print("  queue %s  %3d x %s" % (q, c, n))
print("timeline (duration us / gap in front us):")
for e, g in zip(step, gaps):
    print("  %7.1f  gap %6.2f  %s" % ((e[1] - e[0]) / 1e3, g, short(e[2])))
PY
rm -rf $out
head -60 gpurun_out/timeline.txt
