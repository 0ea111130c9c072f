cd $GRAFT_REPO_ROOT
timeout 2400 python -m pytest tests -x -q -m gpu 2>&1 | tail -6 > gpurun_out/r02_gpu_tests.txt
timeout 300 python tools/in_sweep.py > gpurun_out/r02_in_sweep.md 2> gpurun_out/r02_in_sweep.err
head -5 gpurun_out/r02_gpu_tests.txt; cat gpurun_out/r02_in_sweep.md; tail -3 gpurun_out/r02_in_sweep.err
