#!/bin/bash
# usage: tools/pmc.sh <tag> <k1 args...> ; collects PMC passes for the kernel run by tools/k1.py
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
tag=$1; shift
out=gpurun_out/pmc_$tag; mkdir -p $out
python tools/k1.py "$@" > /dev/null 2>&1   # warm the batch cache
i=0
for set in "SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_VMEM_RD" \
           "SQ_INSTS_LDS SQ_INSTS_SALU SQ_INSTS_MFMA SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_LDS SQ_INST_LEVEL_VMEM SQ_LDS_BANK_CONFLICT SQ_INSTS_VMEM_WR" "SQ_LDS_IDX_ACTIVE SQ_LDS_BANK_CONFLICT SQ_WAIT_INST_LDS SQ_LDS_ADDR_CONFLICT SQ_LDS_DATA_FIFO_FULL SQ_LDS_CMD_FIFO_FULL SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU_MFMA_MOPS_BF16" \
           "TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum TCC_EA0_RDREQ_sum" \
           "FETCH_SIZE" "WRITE_SIZE TCC_EA0_WRREQ_sum" \
           "TCP_PERF_SEL_TOTAL_READ TCP_PERF_SEL_TOTAL_HIT_LRU_READ TCP_PENDING_STALL_CYCLES_sum TCP_READ_TAGCONFLICT_STALL_CYCLES_sum" \
           "GRBM_GUI_ACTIVE GRBM_TA_BUSY"; do
  i=$((i+1))
  timeout 90 rocprofv3 --kernel-trace --pmc $set -f csv -d $out -o p$i -- python tools/k1.py "$@" > $out/log$i.txt 2>&1
done
python - <<PY
import csv, glob, collections
agg = collections.defaultdict(lambda: collections.defaultdict(list))
for f in sorted(glob.glob("$out/*counter_collection.csv")):
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"]
        if "pack_weights" in k or "at::" in k or "elementwise" in k: continue
        agg[k[:70]][r["Counter_Name"]].append(float(r["Counter_Value"]))
with open("$out/summary.txt", "w") as o:
    for k, d in agg.items():
        o.write(k + "\n")
        for c, v in sorted(d.items()):
            o.write("   %-45s n=%3d mean=%.4g\n" % (c, len(v), sum(v) / len(v)))
print(open("$out/summary.txt").read())
PY
rm -f $out/*kernel_trace.csv $out/*agent_info.csv
