#!/usr/bin/env python
"""Per-step table of a rocprofv3 --kernel-trace --stats summary (k_kernel_stats.csv) of a bench.py run:
kernel family totals and the top kernels.  usage: kstats.py <csv> <steps incl. warm-up> [top]"""
import collections, csv, re, sys
rows = list(csv.DictReader(open(sys.argv[1])))
steps = float(sys.argv[2]); top = int(sys.argv[3]) if len(sys.argv) > 3 else 40
FAM = [("conv_tile", "conv_tile"), ("conv_fast", "conv_fast"), ("conv_wlds", "conv_wlds"), ("wgrad", "wgrad"), ("bn", "bn_"),
       ("rulebook", "subm_|down2_|pairs_|scan_|tilebook|conv_assign|conv_tables"), ("pack", "pack_weights"), ("voxel", "voxel"),
       ("ce/sgd/glue", "ce_|sgd|cast_colsum|pad_channels"), ("fill/copy", "fillBuffer|copyBuffer"), ("torch", "at::")]
def family(n):
    for k, pat in FAM:
        if re.search(pat, n): return k
    return "other"
agg = collections.defaultdict(lambda: [0.0, 0.0])
for r in rows:
    if "spin_kernel" in r["Name"]: continue
    f = family(r["Name"]); agg[f][0] += float(r["Calls"]) / steps; agg[f][1] += float(r["TotalDurationNs"]) / 1e3 / steps
tl = sum(v[0] for v in agg.values()); tt = sum(v[1] for v in agg.values())
print("per step: %.1f launches, %.1f us of kernels" % (tl, tt))
for k, v in sorted(agg.items(), key=lambda kv: -kv[1][1]):
    print("  %-12s %6.1f launches %8.1f us" % (k, v[0], v[1]))
for r in sorted(rows, key=lambda r: -float(r["TotalDurationNs"]))[:top]:
    if "spin_kernel" in r["Name"]: continue
    name = re.sub(r"\(anonymous namespace\)::", "", r["Name"])
    print("  %-86s %5.1f x %7.1f us = %7.1f" % (name[:86], float(r["Calls"]) / steps, float(r["AverageNs"]) / 1e3, float(r["TotalDurationNs"]) / 1e3 / steps))
