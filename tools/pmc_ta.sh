#!/bin/bash
# usage: tools/pmc_ta.sh <tag> <k1 args...> : texture-path (TA / TCP / TD) counters of the kernel tools/k1.py runs
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
tag=$1; shift
out=gpurun_out/pmcta_$tag; rm -rf $out; mkdir -p $out
python tools/k1.py "$@" > /dev/null 2>&1
i=0
for set in "GRBM_GUI_ACTIVE TA_TA_BUSY_sum TA_BUFFER_TOTAL_CYCLES_sum TA_BUFFER_WAVEFRONTS_sum" \
           "TA_ADDR_STALLED_BY_TC_CYCLES_sum TA_ADDR_STALLED_BY_TD_CYCLES_sum TA_DATA_STALLED_BY_TC_CYCLES_sum TA_BUFFER_COALESCED_READ_CYCLES_sum" \
           "TCP_GATE_EN1_sum TCP_GATE_EN2_sum TCP_TCP_TA_DATA_STALL_CYCLES_sum TCP_TD_TCP_STALL_CYCLES_sum" \
           "TCP_TCR_TCP_STALL_CYCLES_sum TCP_PENDING_STALL_CYCLES_sum TCP_READ_TAGCONFLICT_STALL_CYCLES_sum TCP_LFIFO_STALL_CYCLES_sum" \
           "TCP_TOTAL_CACHE_ACCESSES_sum TCP_TOTAL_ACCESSES_sum TCP_TCC_READ_REQ_sum TCP_TAGRAM0_REQ_sum" \
           "TCP_TCC_READ_REQ_LATENCY_sum TCP_TCP_LATENCY_sum TCP_RFIFO_STALL_CYCLES_sum TCP_UTCL1_TRANSLATION_MISS_sum" \
           "TD_TD_BUSY_sum TD_TC_STALL_sum TD_SPI_STALL_sum TD_LOAD_WAVEFRONT_sum"; do
  i=$((i+1))
  timeout 90 rocprofv3 --kernel-trace --pmc $set -f csv -d $out -o p$i -- python tools/k1.py "$@" > $out/log$i.txt 2>&1
done
python - <<PY
import csv, glob, collections
agg = collections.defaultdict(lambda: collections.defaultdict(list))
for f in sorted(glob.glob("$out/*counter_collection.csv")):
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"]
        if "conv_fast" not in k and "wgrad" not in k: continue
        agg[k[:70]][r["Counter_Name"]].append(float(r["Counter_Value"]))
with open("$out/summary.txt", "w") as o:
    for k, d in agg.items():
        o.write(k + "\n")
        for c, v in sorted(d.items()):
            o.write("   %-45s n=%3d mean=%.4g\n" % (c, len(v), sum(v) / len(v)))
print(open("$out/summary.txt").read())
PY
rm -f $out/*.csv
