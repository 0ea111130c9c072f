#!/bin/bash
# the InstanceNorm-backward rows of conv1's data gradient finalised inside the block's slab-reduction launch (RSUPER_FUSE_STATS_REDUCE=1, default) vs their own launch
python -m pytest tests/test_gpu_parity.py tests/test_gpu_fullsize.py -q -x -k "basic_block or unet_tiny or two_source or train_steps or determin or reproducible" 2>&1 | tail -3
for i in 1 2 3; do for f in 1 0; do RSUPER_FUSE_STATS_REDUCE=$f python bench.py --no-cpu-baseline --no-secondary --steps 30 --warmup 8 2>/dev/null | python -c "
import json,sys;d=json.loads(sys.stdin.read().strip().splitlines()[-1]);r=d['roofline'];print('FUSE=$f', round(d['ms_per_step'],3), 'frac', round(r['frac'],4), 'conv', round(r['conv_ms_per_step'],3), 'loss', d['config']['final_loss'])"; done; done
