cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r2c
timeout 900 python -m pytest tests/test_gpu_round2.py -q -m gpu -x > gpurun_out/r2c/t_round2.log 2>&1; echo "round2 tests rc=$?"
tail -12 gpurun_out/r2c/t_round2.log
timeout 300 python tools/kbench.py --levels 1,2,3,4,5 --dtypes bf16 > gpurun_out/r2c/kbench.log 2>&1; cat gpurun_out/r2c/kbench.log
bash tools/stepprof.sh r2c
DODA_WGRAD_PAIRS=0 bash tools/stepprof.sh r2c_nopairs
timeout 300 python bench.py --no-cpu-baseline --steps 50 --warmup 15 > gpurun_out/r2c/bench_pairs.json 2>gpurun_out/r2c/bench_pairs.err; tail -c 400 gpurun_out/r2c/bench_pairs.json
DODA_WGRAD_PAIRS=0 timeout 300 python bench.py --no-cpu-baseline --steps 50 --warmup 15 > gpurun_out/r2c/bench_nopairs.json 2>gpurun_out/r2c/bench_nopairs.err; tail -c 400 gpurun_out/r2c/bench_nopairs.json
