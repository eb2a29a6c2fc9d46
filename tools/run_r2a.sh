cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r2a
timeout 900 python -m pytest tests/test_gpu_round2.py -x -q -m gpu > gpurun_out/r2a/t_round2.log 2>&1; echo "round2 tests rc=$?"
tail -15 gpurun_out/r2a/t_round2.log
timeout 300 python tools/kbench.py --levels 1,2,3,4 --dtypes bf16 > gpurun_out/r2a/kbench.log 2>&1; cat gpurun_out/r2a/kbench.log
timeout 300 python tools/kbench.py --scenes 1 --levels 1 --dtypes bf16 >> gpurun_out/r2a/kbench.log 2>&1; tail -1 gpurun_out/r2a/kbench.log
bash tools/stepprof.sh r2a
