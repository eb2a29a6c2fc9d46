cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r2f
timeout 60 python tools/hostprof.py 20000 > /dev/null 2>&1   # warm the box
for cfg in "DODA_WGRAD_PAIRS=1 DODA_BN_FUSION=1" "DODA_WGRAD_PAIRS=0 DODA_BN_FUSION=1" "DODA_WGRAD_PAIRS=1 DODA_BN_FUSION=0" "DODA_WGRAD_PAIRS=0 DODA_BN_FUSION=0" "DODA_WGRAD_PAIRS=1 DODA_BN_FUSION=1"; do
  echo "== $cfg"
  env $cfg timeout 300 python tools/hostprof.py 150000 2>&1 | grep -E "host issue" 
done
timeout 900 python -m pytest tests/test_gpu_round2.py -q -m gpu -x > gpurun_out/r2f/t.log 2>&1; echo "tests rc=$?"; tail -3 gpurun_out/r2f/t.log
bash tools/stepprof.sh r2f
