cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r2b
timeout 300 python tools/graddiag.py > gpurun_out/r2b/graddiag.log 2>&1; tail -15 gpurun_out/r2b/graddiag.log
timeout 900 python -m pytest tests/test_gpu_round2.py -q -m gpu > gpurun_out/r2b/t_round2.log 2>&1; echo "round2 tests rc=$?"
tail -25 gpurun_out/r2b/t_round2.log
timeout 300 python tools/kbench.py --levels 1,2,3,4,5 --dtypes bf16 > gpurun_out/r2b/kbench.log 2>&1; cat gpurun_out/r2b/kbench.log
bash tools/stepprof.sh r2b
