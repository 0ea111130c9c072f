cd $GRAFT_REPO_ROOT
timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_edge.py -x -q 2>&1 | grep -E "passed|failed|Error" | tail -3
bash tools/ab.sh 2>&1 | grep -v amdgpu.ids
