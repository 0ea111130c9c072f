#!/bin/bash
# up4.0's fused [y1 | shortcut] output (32 + 32 bf16 channels) as two tensors (RSUPER_SPLIT_YS=1, default) vs one interleaved 64-channel tensor
python -m pytest tests -m gpu -q -x -k "basic_block or two_source or unet_tiny or fullsize_f32 or train_steps or determin or graphed_step" 2>&1 | tail -2
for i in 1 2 3; do for f in 1 0; do RSUPER_SPLIT_YS=$f python bench.py --no-cpu-baseline --no-secondary --steps 30 --warmup 8 2>/dev/null | python -c "
import json,sys;d=json.loads(sys.stdin.read().strip().splitlines()[-1]);r=d['roofline'];print('SPLIT_YS=$f', round(d['ms_per_step'],3), 'frac', round(r['frac'],4), 'conv', round(r['conv_ms_per_step'],3), 'loss', d['config']['final_loss'])"; done; done
python tools/in_bwd_rates.py 2>/dev/null | sed -n 3,8p
