cd $GRAFT_REPO_ROOT
timeout 1500 python -m pytest tests -x -q -m gpu 2>&1 | tail -6 > gpurun_out/r02_gpu_tests.txt
timeout 600 python tools/bench_conv.py bf16 > gpurun_out/r02_conv_table_ws4.txt 2>&1
timeout 900 python bench.py > gpurun_out/r02_bench_a.json 2> gpurun_out/r02_bench_a.err
tail -5 gpurun_out/r02_gpu_tests.txt; cat gpurun_out/r02_conv_table_ws4.txt; tail -c 3000 gpurun_out/r02_bench_a.json; tail -5 gpurun_out/r02_bench_a.err
