cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r2j
timeout 2400 python -m pytest tests -q -m gpu -x > gpurun_out/r2j/t.log 2>&1; echo "gpu tests rc=$?"; tail -25 gpurun_out/r2j/t.log
