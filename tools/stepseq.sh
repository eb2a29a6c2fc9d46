#!/bin/bash
# One training step as the ordered list of kernels per queue (rocprofv3 kernel trace of a short bench): what is launched, where.
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
out=gpurun_out/seq; rm -rf $out; mkdir -p $out
timeout 300 rocprofv3 --kernel-trace -f csv -d $out -o k -- python bench.py --no-cpu-baseline --no-train-entry --config5-steps 0 --fp32-steps 0 --kernel-reps 0 --steps 12 --warmup 8 > $out/bench.json 2> $out/err.txt
python - <<PY
import csv, collections, re
rows = list(csv.DictReader(open("$out/k_kernel_trace.csv")))
qs = collections.Counter(r["Queue_Id"] for r in rows)
main = qs.most_common(1)[0][0]
ev = sorted(((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"], r["Queue_Id"], r.get("Grid_Size", r.get("Grid_Size_X", ""))) for r in rows))
sgd = [k for k, e in enumerate(ev) if "sgd" in e[2]]
a, b = sgd[-3], sgd[-2]
def short(n):
    n = re.sub(r"\(anonymous namespace\)::", "", n)
    n = re.sub(r"^void ", "", n)
    n = n.split("(")[0]
    return n[:70]
seq = [(short(e[2]), e[3] == main, (e[1] - e[0]) / 1e3) for e in ev[a + 1:b + 1]]
print("queues:", dict(qs), "main:", main, " kernels in the step:", len(seq), " on the main queue:", sum(1 for s in seq if s[1]))
out = []
for name, on_main, us in seq:
    tag = name if on_main else "  [side] " + name
    if out and out[-1][0] == tag:
        out[-1][1] += 1; out[-1][2] += us
    else:
        out.append([tag, 1, us])
for tag, n, us in out:
    print("%3d x %7.1f us  %s" % (n, us, tag))
PY
rm -f $out/*.csv
