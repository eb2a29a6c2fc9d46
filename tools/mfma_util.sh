#!/bin/bash
# MFMA utilisation (SQ_VALU_MFMA_BUSY_CYCLES against GRBM_GUI_ACTIVE x SIMDs) of the level-1 conv / wgrad kernels
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
out=gpurun_out/mfma; rm -rf $out; mkdir -p $out
python tools/k1.py bf16 fwd 16 > /dev/null 2>&1
for which in fwd wgrad; do
  timeout 90 rocprofv3 --kernel-trace --pmc SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_BUSY_CYCLES -f csv -d $out/$which -o p -- python tools/k1.py bf16 $which 16 > $out/$which.log 2>&1
done
python - <<PY
import csv, glob, collections
for which in ("fwd", "wgrad"):
    agg = collections.defaultdict(lambda: collections.defaultdict(list))
    for f in glob.glob("$out/%s/*counter_collection.csv" % which):
        for r in csv.DictReader(open(f)):
            k = r["Kernel_Name"]
            if "conv_fast" in k or "wgrad_kernel" in k:
                agg[k[:60]][r["Counter_Name"]].append(float(r["Counter_Value"]))
    for k, d in agg.items():
        m = {c: sum(v) / len(v) for c, v in d.items()}
        line = "%s: " % k + ", ".join("%s=%.4g" % kv for kv in sorted(m.items()))
        if "SQ_VALU_MFMA_BUSY_CYCLES" in m and "GRBM_GUI_ACTIVE" in m:
            line += "  | MFMA busy / (GUI_ACTIVE per XCD x 1024 SIMDs) = %.2f %%" % (100.0 * m["SQ_VALU_MFMA_BUSY_CYCLES"] / (m["GRBM_GUI_ACTIVE"] / 8.0 * 1024.0))
        print(line)
PY
rm -rf $out/fwd $out/wgrad
