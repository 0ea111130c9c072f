# LDS-DMA weight gradient: full kernel / MFMA loop alone (no DMA in the loop) / DMA stream alone (no MFMAs); kernel durations from a kernel trace
# usage: bash tools/wgd_ablate.sh "<layer> wgrad xhat" ...
cd $GRAFT_REPO_ROOT
: > gpurun_out/wgd_ablate.txt
for FL in "" "-DWGD_SKIP_DMA" "-DWGD_SKIP_MMA" $WGD_FLAGS; do
  (cd r-super_amd/csrc && rm -f _build/conv3d_wgrad_dma.o && make WGD_EXTRA="$FL" > /dev/null 2>&1)
  echo "#### build [$FL]" >> gpurun_out/wgd_ablate.txt
  bash tools/kt_one.sh "$@"; grep "==\|wgrad_dma" gpurun_out/kt_one.txt | grep -v reduce >> gpurun_out/wgd_ablate.txt
done
