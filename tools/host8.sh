#!/bin/bash
# Host issue time per step with 1 and with 8 training processes running concurrently on one host: tools/hostab.py
# on scenes small enough that the GPU has nothing to do (the processes share ONE GPU here; 8 ranks of a node share the host the same way).
# usage: tools/host8.sh [voxels per scene = 2000]   -> gpurun_out/host8.txt
cd ${GRAFT_REPO_ROOT:-/root/repo}
v=${1:-2000}; out=gpurun_out/host8.txt; : > $out
echo "== 1 process" >> $out
python tools/hostab.py $v 60 3 2>&1 | grep "blocks=1" >> $out
for n in 4 8; do
  echo "== $n processes" >> $out
  for i in $(seq 1 $n); do python tools/hostab.py $v 60 3 > /tmp/host8_$i.log 2>&1 & done
  wait
  for i in $(seq 1 $n); do grep "blocks=1" /tmp/host8_$i.log >> $out; done
done
nproc >> $out; uptime >> $out
cat $out
