cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r2g
timeout 60 python tools/hostprof.py 20000 > /dev/null 2>&1   # warm the box
for cfg in "PREFETCH=1 DODA_WGRAD_PAIRS=1" "PREFETCH=0 DODA_WGRAD_PAIRS=1" "PREFETCH=1 DODA_WGRAD_PAIRS=0" "PREFETCH=1 DODA_WGRAD_PAIRS=1 DODA_BN_FUSION=0" "PREFETCH=1 DODA_WGRAD_PAIRS=1"; do
  echo "== $cfg"
  env $cfg timeout 300 python tools/hostprof.py 150000 2>&1 | grep -E "host issue" 
done
timeout 900 python -m pytest tests/test_gpu_round2.py tests/test_gpu_unet.py -q -m gpu -x > gpurun_out/r2g/t.log 2>&1; echo "tests rc=$?"; tail -3 gpurun_out/r2g/t.log
timeout 300 python bench.py --no-cpu-baseline --steps 60 --warmup 20 > gpurun_out/r2g/bench.json 2>gpurun_out/r2g/bench.err; python -c "
import json; d=json.loads(open('gpurun_out/r2g/bench.json').read().strip().splitlines()[-1]); print('ms/step', d['ms_per_step'], 'loss', d['config']['final_loss'])"
timeout 300 python bench.py --no-cpu-baseline --steps 60 --warmup 20 --prefetch 0 > gpurun_out/r2g/bench_nopf.json 2>gpurun_out/r2g/bench_nopf.err; python -c "
import json; d=json.loads(open('gpurun_out/r2g/bench_nopf.json').read().strip().splitlines()[-1]); print('noprefetch ms/step', d['ms_per_step'], 'loss', d['config']['final_loss'])"
