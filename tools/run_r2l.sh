cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
timeout 600 python tools/abbench.py 40
timeout 600 python -m pytest tests/test_gpu_round2.py -q -m gpu -x -k "epilogue or bn_fusion" 2>&1 | tail -2
