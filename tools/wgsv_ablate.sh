# small-volume weight gradient: full / no staging / no MFMA loop (kernel durations from a kernel trace)
cd $GRAFT_REPO_ROOT
: > gpurun_out/wgsv_ablate.txt
for FL in "" "-DWGSV_SKIP_STAGE" "-DWGSV_SKIP_MMA" "-DWGSV_SKIP_STORE" "-DWGSV_SKIP_STAGE -DWGSV_SKIP_MMA -DWGSV_SKIP_STORE"; do
  (cd r-super_amd/csrc && rm -f _build/conv3d_wgrad_sv.o && make WGSV_EXTRA="$FL" > /dev/null 2>&1)
  echo "#### build [$FL]" >> gpurun_out/wgsv_ablate.txt
  bash tools/kt_one.sh "7 wgrad" "8 wgrad"; grep "==\|wgrad_sv" gpurun_out/kt_one.txt | cut -c1-50,98-140 >> gpurun_out/wgsv_ablate.txt
done
(cd r-super_amd/csrc && rm -f _build/conv3d_wgrad_sv.o && make > /dev/null 2>&1)
