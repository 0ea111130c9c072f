#!/bin/bash
# data gradient of the fused [conv1 | shortcut] of down3.0 (512 rows of dY -> 128 columns at 12^3: 108 blocks of the volume-fitted kernel) under other kernels
L='down3.0 128->256+sc:12:128:0:256:1;down2.0 64->128+sc:24:64:0:128:1;down4.0 256->320+sc:6:256:0:320:1;up1.0 576->256+sc:12:256:320:256:1'
for cfg in "X=1" "RSUPER_NO_BOX=1" "RSUPER_IGEMM_VARIANT=1" "RSUPER_IGEMM_VARIANT=0"; do echo "== $cfg"; env $cfg BC_LAYERS="$L" python tools/bench_conv.py bf16 2>&1 | grep -v "^#\|amdgpu.ids" | cut -c1-210; done
