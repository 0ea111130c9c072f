bash tools/wg_mt1_exp.sh 2>&1 | tail -50
for f in 0 1 0 1; do RSUPER_TIMER_FENCED=$f python bench.py --no-cpu-baseline --no-secondary --steps 30 --warmup 8 2>/dev/null | python -c "
import json,sys;d=json.loads(sys.stdin.read().strip().splitlines()[-1]);r=d['roofline'];print('fenced=$f', round(d['ms_per_step'],3), 'frac', round(r['frac'],4), 'conv_ms', round(r['conv_ms_per_step'],3), 'sum', round(r['conv_ms_per_step_sum_of_launches'],3), {k:round(v['avg_us'],1) for k,v in r['per_kernel'].items()})"; done
