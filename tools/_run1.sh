cd $GRAFT_REPO_ROOT
timeout 300 python tools/in_sweep.py 2>/dev/null | grep -E "upsample|maxpool" | head -8
timeout 600 python -m pytest tests/test_gpu_parity.py -x -q -m gpu 2>&1 | grep -E "passed|failed" | tail -2
