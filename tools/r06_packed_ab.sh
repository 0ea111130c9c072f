python -m pytest tests/test_gpu_edge.py -q -x -k "packed or hip_graph" 2>&1 | tail -5
python tools/run_exact.py conv_exact_s2 2>&1 | tail -25
for i in 1 2; do
python bench.py --no-cpu-baseline --no-secondary --steps 30 --warmup 8 > gpurun_out/b_packed_$i.json 2>gpurun_out/b_packed.err; python -c "
import json;d=json.loads(open('gpurun_out/b_packed_$i.json').read().strip().splitlines()[-1]);print('packed', d['ms_per_step'], d['roofline']['frac'], d['config']['final_loss'])"
python bench.py --no-cpu-baseline --no-secondary --steps 30 --warmup 8 --unpacked-labels > gpurun_out/b_unpacked_$i.json 2>gpurun_out/b_unpacked.err; python -c "
import json;d=json.loads(open('gpurun_out/b_unpacked_$i.json').read().strip().splitlines()[-1]);print('unpacked', d['ms_per_step'], d['roofline']['frac'], d['config']['final_loss'])"
done
python bench.py --no-cpu-baseline --no-secondary --steps 20 --warmup 5 --report > gpurun_out/b_c3_packed.json 2>gpurun_out/b_c3.err; python -c "
import json;d=json.loads(open('gpurun_out/b_c3_packed.json').read().strip().splitlines()[-1]);print('c3 packed', d['ms_per_step'], d['config']['final_loss'])"
python bench.py --no-cpu-baseline --no-secondary --steps 20 --warmup 5 --report --unpacked-labels > gpurun_out/b_c3_unpacked.json 2>gpurun_out/b_c3u.err; python -c "
import json;d=json.loads(open('gpurun_out/b_c3_unpacked.json').read().strip().splitlines()[-1]);print('c3 unpacked', d['ms_per_step'], d['config']['final_loss'])"
tail -3 gpurun_out/b_packed.err gpurun_out/b_c3.err
python -m pytest tests/test_gpu_fullsize.py -q -x -k "config2_fullsize_f32" -s 2>&1 | grep -v "^oracle" | cut -c1-420 | head -30
