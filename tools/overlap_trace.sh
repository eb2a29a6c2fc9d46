#!/bin/bash
# usage: tools/overlap_trace.sh <tag>: kernel trace of the bench step under a forced one-rank RCCL process group; prints where the
# all-reduce kernels run relative to the step's own kernels (tools/overlap_trace.py) -> gpurun_out/ov_<tag>/summary.txt
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
tag=${1:-x}; out=gpurun_out/ov_$tag; rm -rf $out; mkdir -p $out
export DODA_DIST_FORCE=1 DODA_DIST_BACKEND=nccl HSA_ENABLE_IPC_MODE_LEGACY=0
timeout 600 rocprofv3 --kernel-trace -f csv -d $out -o k -- python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29655 \
    bench.py --gpus 1 --no-cpu-baseline --no-train-entry --fp32-steps 0 --kernel-reps 1 --steps 12 --warmup 4 > $out/bench.json 2> $out/err.txt
f=$(ls $out/*kernel_trace.csv 2>/dev/null | head -1)
python tools/overlap_trace.py "$f" > $out/summary.txt 2>&1
rm -f $out/*kernel_trace.csv $out/*agent_info.csv
cat $out/summary.txt
