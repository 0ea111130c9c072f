#!/bin/bash
# the fused [conv1 | shortcut] weight gradients of the low-resolution blocks: small-volume kernel (default) vs the tile-streaming kernels
L='up1.0 576->256+sc:12:256:320:256:1;down3.0 128->256+sc:12:128:0:256:1;256->256:12:256:0:256:0;down4.0 256->320+sc:6:256:0:320:1;320->320:6:320:0:320:0;up2.0 384->128+sc:24:128:256:128:1;down2.0 64->128+sc:24:64:0:128:1'
for cfg in "X=1" "RSUPER_WGRAD_SV=0" "RSUPER_WGRAD_SV=0 RSUPER_WGRAD2_MIN_TILES=1"; do echo "== $cfg"; env $cfg BC_LAYERS="$L" python tools/bench_conv.py bf16 2>&1 | grep -v "^#\|amdgpu.ids" | cut -c1-210; done
