cd $GRAFT_REPO_ROOT
python tools/host_time.py 2>&1 | grep "host enqueue" | head -4
bash tools/ab.sh 2>&1 | grep -v amdgpu.ids | cut -c1-20
timeout 600 python -m pytest tests/test_gpu_parity.py -x -q 2>&1 | grep -E "passed|failed" | tail -1
