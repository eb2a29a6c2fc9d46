# conv_tile16 in the step form with statistics as rows / as fp64 totals: average kernel time back to back (rocprofv3).
# (Historical third arm: the packed totals layout, a variant library built with the old stats_emit — 26.4 us against 24.5 rows /
# 24.2 padded; the padded layout is the only one now.)
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
python tools/k1.py bf16 fwd 16 2 > /dev/null 2>&1
for w in fwdstep fwdtot fwdstep fwdtot; do
rm -rf /tmp/totk
if [ $w = pad ]; then K1_LIB=$GRAFT_REPO_ROOT/tools/_totpad/libdoda_hip.so timeout 200 rocprofv3 --kernel-trace --stats -f csv -d /tmp/totk -o k -- python tools/k1.py bf16 fwdtot 16 200 > /dev/null 2>&1
else timeout 200 rocprofv3 --kernel-trace --stats -f csv -d /tmp/totk -o k -- python tools/k1.py bf16 $w 16 200 > /dev/null 2>&1; fi
python - <<PY
import csv
for r in csv.DictReader(open("/tmp/totk/k_kernel_stats.csv")):
    if "conv_tile16" in r["Name"]: print("$w", r["Calls"], "avg us %.2f" % (float(r["AverageNs"]) / 1e3), "min %.2f" % (float(r["MinNs"]) / 1e3))
PY
done
