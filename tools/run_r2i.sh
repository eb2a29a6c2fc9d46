cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r2i
timeout 1500 python -m pytest tests/test_harness.py tests/test_gpu_dist.py -q -m gpu -x > gpurun_out/r2i/t.log 2>&1; echo "tests rc=$?"; tail -30 gpurun_out/r2i/t.log
