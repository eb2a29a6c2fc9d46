cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
cat > /tmp/vp.py <<'PY'
import sys, torch
sys.path.insert(0, "/root/repo")
from doda_amd import ops
from doda_amd.scene import make_batch
b = make_batch(4, 150000, 1000)
locs = b["locs"].to("cuda:0")
m, w = b["v2p_map"].shape
for _ in range(60):
    ops.voxelize_idx_device(locs, 4, 4, sizes=(m, w - 1))
torch.cuda.synchronize()
PY
rm -rf /tmp/vp; timeout 300 rocprofv3 --kernel-trace --stats -f csv -d /tmp/vp -o k -- python /tmp/vp.py > /dev/null 2>&1
python - <<PY
import csv
tot = 0
for r in csv.DictReader(open("/tmp/vp/k_kernel_stats.csv")):
    c = int(r["Calls"])
    if c >= 60:
        print("%-60s %4d x %7.2f us = %7.1f us per call" % (r["Name"].replace("(anonymous namespace)::", "")[:60], c // 60, float(r["AverageNs"]) / 1e3, c / 60 * float(r["AverageNs"]) / 1e3))
        tot += c / 60 * float(r["AverageNs"]) / 1e3
print("sum per call: %.1f us" % tot)
PY
