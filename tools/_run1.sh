cd $GRAFT_REPO_ROOT
timeout 1500 python -m pytest tests -x -q -m gpu 2>&1 | tail -6 | head -2 > gpurun_out/r02_gpu_tests.txt
timeout 900 python bench.py --no-cpu-baseline > gpurun_out/r02_bench_b.json 2> gpurun_out/r02_bench_b.err
cat gpurun_out/r02_gpu_tests.txt; python -c "
import json; d=json.loads(open('gpurun_out/r02_bench_b.json').read().strip().splitlines()[-1]); print(d['ms_per_step'], d['roofline']['frac'], d['roofline']['conv_ms_per_step'], {k:(round(v['avg_us'],1)) for k,v in d['roofline']['per_kernel'].items()}, d.get('secondary'))"
