cd $GRAFT_REPO_ROOT
timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_fuzz.py tests/test_gpu_edge.py -x -q -m gpu 2>&1 | tail -4 > gpurun_out/r02_ws_tests.txt
timeout 600 python tools/bench_conv.py bf16 2>&1 | grep -v amdgpu | cut -c1-60,118-160 > gpurun_out/r02_conv_table_w3.txt
tail -3 gpurun_out/r02_ws_tests.txt; cat gpurun_out/r02_conv_table_w3.txt
