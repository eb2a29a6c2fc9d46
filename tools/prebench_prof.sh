#!/bin/bash
# kernel durations (rocprofv3) of tools/prebench.py per DODA_PRE_ABLATE value: gpurun_out/prebench_<a>.txt
cd /tmp && export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-/root/repo}
for a in ${ABL:-0 7}; do
  rm -rf /tmp/pb_$a
  DODA_PRE_ABLATE=$a rocprofv3 --kernel-trace --stats -f csv -d /tmp/pb_$a -o k -- python $R/tools/prebench.py ${CFG:-1900 80} > /tmp/pb_$a.log 2>&1
  f=$(find /tmp/pb_$a -name "*kernel_stats.csv" | head -1)
  echo "ablate $a"; grep rows /tmp/pb_$a.log
  python - "$f" <<'PY'
import csv, re, sys
rows = list(csv.DictReader(open(sys.argv[1])))
for r in sorted(rows, key=lambda r: -float(r["TotalDurationNs"]))[:12]:
    n = re.sub(r"\(anonymous namespace\)::", "", r["Name"])[:100]
    print("  %-100s calls %6s avg %7.2f us" % (n, r["Calls"], float(r["AverageNs"]) / 1e3))
PY
done
