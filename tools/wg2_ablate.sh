# second-generation weight gradient: full kernel / MFMA loop alone / staging alone; kernel durations from a kernel trace
# usage: bash tools/wg2_ablate.sh "<layer> wgrad" ...     extra builds: WG2_FLAGS="-DX -DY"
cd $GRAFT_REPO_ROOT
: > gpurun_out/wg2_ablate.txt
for FL in "" "-DWG2_SKIP_STAGE" "-DWG2_SKIP_MMA" $WG2_FLAGS; do
  (cd r-super_amd/csrc && rm -f _build/conv3d_wgrad2.o && make WG2_EXTRA="$FL" > /dev/null 2>&1)
  echo "#### build [$FL]" >> gpurun_out/wg2_ablate.txt
  bash tools/kt_one.sh "$@"; grep "==\|wgrad2" gpurun_out/kt_one.txt | grep -v reduce | cut -c1-45,100-140 >> gpurun_out/wg2_ablate.txt
done
