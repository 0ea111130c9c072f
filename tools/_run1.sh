cd $GRAFT_REPO_ROOT
timeout 900 python -m pytest tests/test_gpu_edge.py -x -q -m gpu -k "driver or self_spawn" 2>&1 | grep -E "passed|failed|Error|assert" | tail -8
