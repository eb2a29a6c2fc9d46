cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r2e
for cfg in "DODA_WGRAD_PAIRS=1 DODA_BN_FUSION=1" "DODA_WGRAD_PAIRS=0 DODA_BN_FUSION=1" "DODA_WGRAD_PAIRS=1 DODA_BN_FUSION=0" "DODA_WGRAD_PAIRS=0 DODA_BN_FUSION=0"; do
  echo "== $cfg"
  env $cfg timeout 300 python tools/hostprof.py 150000 2>&1 | grep -E "host issue|deferred" 
done
env DODA_WGRAD_PAIRS=1 DODA_BN_FUSION=1 timeout 300 python tools/hostprof.py 150000 > gpurun_out/r2e/hostprof_full.log 2>&1
