#!/bin/bash
# Weight gradient of the mid-resolution layers: 64-row blocks of the first-generation kernel (32 K splits) against 32-row blocks of the second-generation
# kernel (half the splits / slab bytes, twice the tiles per block).  One process per configuration (the switches are read once).
L='64->64,down1.0,128->128,up2.0,up3.0,256->256'
run() { echo "== $*"; env "$@" BC_ONLY="$L" python tools/bench_conv.py bf16 2>&1 | grep -v '^#' | sed -E 's/\| fwd.*\| wgrad/| wgrad/'; }
run BASE=1
run RSUPER_WG2_MT1_MAXM=256 RSUPER_WG2_MT1_MIN_TILES=128 RSUPER_WGRAD2_MIN_TILES=8
run RSUPER_WG2_MT1_MAXM=256 RSUPER_WG2_MT1_MIN_TILES=128 RSUPER_WGRAD2_MIN_TILES=4
run RSUPER_WG2_MT1_MAXM=128 RSUPER_WG2_MT1_MIN_TILES=128 RSUPER_WGRAD2_MIN_TILES=8
run RSUPER_WG2_MT1=0 RSUPER_WGRAD2_MIN_TILES=4
run RSUPER_WG2_MT1=0 RSUPER_WGRAD2_MIN_TILES=2
