#!/bin/bash
# strided weight gradient: where does the tile time go (profiling switches of conv3d_wgrad_s2.hip; rebuilt on the box)
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out; out=gpurun_out/s2w_variants.txt; : > $out
for v in ${S2W_VARIANTS:-"-DS2W_SKIP=0" "-DS2W_SKIP=15" "-DS2W_SKIP=31" "-DS2W_SKIP=32" "-DS2W_SKIP=48"}; do
  touch r-super_amd/csrc/conv3d_wgrad_s2.hip
  timeout 200 make -C r-super_amd/csrc S2W_EXTRA="$v" > /dev/null 2>&1 || { echo "build failed $v" >> $out; continue; }
  echo "== $v" >> $out
  BC_ONLY_S2=1 timeout 200 python tools/bench_conv.py bf16 2>&1 | grep -o "^down.\.0 s2 [0-9>-]*\|wgrad *[0-9.]* us" | paste - - >> $out
done
touch r-super_amd/csrc/conv3d_wgrad_s2.hip
cat $out
