cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r2d
timeout 1500 python -m pytest tests -q -m gpu -x > gpurun_out/r2d/t_gpu.log 2>&1; echo "gpu tests rc=$?"
tail -15 gpurun_out/r2d/t_gpu.log
bash tools/stepprof.sh r2d
DODA_BN_FUSION=0 bash tools/stepprof.sh r2d_nofuse
timeout 300 python bench.py --no-cpu-baseline --steps 50 --warmup 15 > gpurun_out/r2d/bench.json 2>gpurun_out/r2d/bench.err; tail -c 300 gpurun_out/r2d/bench.json; python -c "
import json; d=json.loads(open('gpurun_out/r2d/bench.json').read().strip().splitlines()[-1]); print('ms/step', d['ms_per_step'], 'loss', d['config']['final_loss'])"
DODA_BN_FUSION=0 timeout 300 python bench.py --no-cpu-baseline --steps 50 --warmup 15 > gpurun_out/r2d/bench_nofuse.json 2>gpurun_out/r2d/bench_nofuse.err; python -c "
import json; d=json.loads(open('gpurun_out/r2d/bench_nofuse.json').read().strip().splitlines()[-1]); print('nofuse ms/step', d['ms_per_step'], 'loss', d['config']['final_loss'])"
