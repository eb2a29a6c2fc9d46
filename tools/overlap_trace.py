"""Where the gradient all-reduce runs inside the step: from a rocprofv3 kernel trace of bench.py under a forced one-rank RCCL
group, per step the span of the RCCL kernels and the step's own kernels that run DURING it."""
import collections, csv, re, sys
rows = list(csv.DictReader(open(sys.argv[1])))
ks = []
for r in rows:
    name = r.get("Kernel_Name") or r.get("Name")
    ks.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), name, r.get("Queue_Id") or r.get("Stream_Id") or "?"))
ks.sort()
is_cc = lambda n: bool(re.search(r"nccl|rccl|AllReduce|msccl", n, re.I))
cc = [k for k in ks if is_cc(k[2])]
print("kernels %d, collective kernels %d" % (len(ks), len(cc)))
if not cc:
    names = collections.Counter(k[2][:60] for k in ks)
    print("no RCCL kernel names found; most frequent kernels:")
    for n, c in names.most_common(12):
        print("  %5d  %s" % (c, n))
    sys.exit(0)
# steps: sgd_multi marks a step's end
ends = [k[1] for k in ks if "sgd_multi" in k[2]]
starts = [ks[0][0]] + ends[:-1]
short = lambda n: re.sub(r"\(anonymous namespace\)::|void ", "", n)[:70]
for si, (a, b) in enumerate(zip(starts, ends)):
    if si < 6:
        continue
    step = [k for k in ks if a <= k[0] < b]
    ccs = [k for k in step if is_cc(k[2])]
    if not ccs:
        continue
    t0 = step[0][0]
    print("step %d: %.2f ms, %d kernels; collectives:" % (si, (b - a) / 1e6, len(step)))
    for c in ccs:
        under = [k for k in step if not is_cc(k[2]) and k[0] < c[1] and k[1] > c[0]]
        fam = collections.Counter(short(k[2]).split("<")[0].split("(")[0] for k in under)
        print("   %-40s %7.1f .. %7.1f us (%.1f us), %d other kernels running meanwhile: %s" % (
            short(c[2])[:40], (c[0] - t0) / 1e3, (c[1] - t0) / 1e3, (c[1] - c[0]) / 1e3, len(under),
            ", ".join("%s x%d" % kv for kv in fam.most_common(5))))
    last_bwd = max((k for k in step if re.search(r"conv_tile|conv_fast|bn_bwd", k[2])), key=lambda k: k[1])
    print("   last conv / BatchNorm-backward kernel of the step ends at %.1f us; step's last kernel at %.1f us" % ((last_bwd[1] - t0) / 1e3, (step[-1][1] - t0) / 1e3))
    if si > 8:
        break
