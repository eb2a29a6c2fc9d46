cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r2k
bash tools/stepprof.sh r2k
for i in 1 2 3; do timeout 300 python bench.py --no-cpu-baseline --steps 60 --warmup 20 --fp32-steps 0 --kernel-reps 2 > gpurun_out/r2k/bench$i.json 2>gpurun_out/r2k/bench.err; python -c "
import json; d=json.loads(open('gpurun_out/r2k/bench$i.json').read().strip().splitlines()[-1]); print('ms/step', d['ms_per_step'], 'loss', d['config']['final_loss'])"; done
DODA_BN_FUSION=0 timeout 300 python bench.py --no-cpu-baseline --steps 60 --warmup 20 --fp32-steps 0 --kernel-reps 2 > gpurun_out/r2k/bench_nf.json 2>gpurun_out/r2k/bench.err; python -c "
import json; d=json.loads(open('gpurun_out/r2k/bench_nf.json').read().strip().splitlines()[-1]); print('nofusion ms/step', d['ms_per_step'])"
DODA_WGRAD_PAIRS=0 timeout 300 python bench.py --no-cpu-baseline --steps 60 --warmup 20 --fp32-steps 0 --kernel-reps 2 > gpurun_out/r2k/bench_np.json 2>gpurun_out/r2k/bench.err; python -c "
import json; d=json.loads(open('gpurun_out/r2k/bench_np.json').read().strip().splitlines()[-1]); print('nopairs ms/step', d['ms_per_step'])"
