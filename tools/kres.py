#!/usr/bin/env python
"""Per-kernel register / LDS / occupancy table of one csrc/*.hip file (hipcc -Rpass-analysis=kernel-resource-usage,
cross-compiled for gfx950: runs without a GPU).  usage: python tools/kres.py spconv_tile [filter]"""
import os
import re
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from doda_amd.build import FLAGS, HIPCC  # noqa: E402

src = os.path.join(ROOT, "doda_amd", "csrc", sys.argv[1] + ("" if sys.argv[1].endswith(".hip") else ".hip"))
flt = sys.argv[2] if len(sys.argv) > 2 else ""
r = subprocess.run([HIPCC, *FLAGS, "-Rpass-analysis=kernel-resource-usage", "-c", src, "-o", "/tmp/_kres.o"],
                   capture_output=True, text=True)
if r.returncode:
    sys.exit(r.stderr[-3000:])
cur, rows = None, []
for line in r.stderr.splitlines():
    m = re.search(r"remark: [^:]+:\d+:\d+:\s+(Function Name|Name): (\S+)", line) or re.search(r"(Function Name|Name): (\S+)", line)
    if m:
        name = subprocess.run(["c++filt", m.group(2)], capture_output=True, text=True).stdout.strip()
        cur = {"name": re.sub(r"\(anonymous namespace\)::", "", name).split("(")[0]}
        rows.append(cur)
        continue
    for key in ("VGPRs", "AGPRs", "TotalSGPRs", "ScratchSize [bytes/lane]", "Occupancy [waves/SIMD]", "LDS Size [bytes/block]"):
        m = re.search(re.escape(key) + r": (\d+)", line)
        if m and cur is not None and key not in cur:
            cur[key] = int(m.group(1))
print("%-64s %5s %5s %5s %7s %4s %7s" % ("kernel", "VGPR", "AGPR", "SGPR", "scratch", "occ", "LDS"))
for c in rows:
    if flt in c["name"]:
        print("%-64s %5s %5s %5s %7s %4s %7s" % (c["name"][-64:], c.get("VGPRs"), c.get("AGPRs"), c.get("TotalSGPRs"),
                                                 c.get("ScratchSize [bytes/lane]"), c.get("Occupancy [waves/SIMD]"),
                                                 c.get("LDS Size [bytes/block]")))
