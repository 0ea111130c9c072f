#!/bin/bash
# up4.0's per-source data gradient into one contiguous tensor per source (RSUPER_SPLIT_G0=1, default) vs column ranges of one 96-channel tensor
python -m pytest tests/test_gpu_fullsize.py -q -x -k "two_source" 2>&1 | tail -2
for i in 1 2 3; do for f in 1 0; do RSUPER_SPLIT_G0=$f python bench.py --no-cpu-baseline --no-secondary --steps 30 --warmup 8 2>/dev/null | python -c "
import json,sys;d=json.loads(sys.stdin.read().strip().splitlines()[-1]);r=d['roofline'];print('SPLIT_G0=$f', round(d['ms_per_step'],3), 'frac', round(r['frac'],4), 'dgrad avg', round(r['per_kernel']['conv3d_igemm_dgrad']['avg_us'],1), 'loss', d['config']['final_loss'])"; done; done
python tools/in_bwd_rates.py 2>/dev/null | sed -n 3,8p
