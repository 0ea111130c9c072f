true
wait
for i in 1 2; do for kd in 0 1; do RSUPER_KD_DGRAD=$kd python bench.py --no-cpu-baseline --no-secondary --steps 30 --warmup 8 2>/dev/null | python -c "
import json,sys;d=json.loads(sys.stdin.read().strip().splitlines()[-1]);r=d['roofline'];print('KD_DGRAD=$kd', round(d['ms_per_step'],3), 'frac', round(r['frac'],4), 'conv', round(r['conv_ms_per_step'],3), {k:round(v['avg_us'],1) for k,v in r['per_kernel'].items()})"; done; done
BC_EXTRA=1 python tools/bench_conv.py bf16 2>&1 | grep -v "^#" | cut -c1-200
BC_EXTRA=1 RSUPER_KD_DGRAD=1 python tools/bench_conv.py bf16 2>&1 | grep -v "^#" | cut -c1-200
