cd $GRAFT_REPO_ROOT
timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_fuzz.py -x -q -m gpu 2>&1 | grep -E "passed|failed|Error|error" | tail -5 > gpurun_out/r02_ws_tests.txt
timeout 600 python tools/bench_conv.py bf16 5 2>&1 | grep -v amdgpu | cut -c1-117 > gpurun_out/r02_conv_table_pc2.txt
cat gpurun_out/r02_ws_tests.txt gpurun_out/r02_conv_table_pc2.txt
