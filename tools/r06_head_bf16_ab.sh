#!/bin/bash
# head backward (weight + bias + data gradient in one pass over dlogits): bf16 MFMA products (default) vs the exact-f32 chains of rounds 4-5
python tools/run_exact.py check_head 2>&1 | tail -12
python -m pytest tests -m gpu -q -x -k "unet_tiny or train_steps or align_with_f32 or head" 2>&1 | tail -3
for i in 1 2 3; do for f in 1 0; do RSUPER_HEAD_BF16=$f python bench.py --no-cpu-baseline --no-secondary --steps 30 --warmup 8 2>/dev/null | python -c "
import json,sys;d=json.loads(sys.stdin.read().strip().splitlines()[-1]);r=d['roofline'];print('HEAD_BF16=$f', round(d['ms_per_step'],3), 'frac', round(r['frac'],4), 'loss', d['config']['final_loss'])"; done; done
RSUPER_HEAD_BF16=1 python tools/bench_small.py 2>&1 | tail -8
RSUPER_HEAD_BF16=0 python tools/bench_small.py 2>&1 | tail -8
