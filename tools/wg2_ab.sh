# A/B of build flags for conv3d_wgrad2.hip on selected layers, same box, interleaved: bash tools/wg2_ab.sh "<flags A>" "<flags B>" ...   (BC_ONLY selects the layers)
cd $GRAFT_REPO_ROOT
: > gpurun_out/wg2_ab.txt
for rep in 1 2; do
for FL in "$@"; do
  (cd r-super_amd/csrc && rm -f _build/conv3d_wgrad2.o && make WG2_EXTRA="$FL" > /dev/null 2>&1)
  echo "#### [$FL] run $rep" >> gpurun_out/wg2_ab.txt
  BC_ONLY=${BC_ONLY:-"inc,up4.0,up3.0,64->64"} python tools/bench_conv.py bf16 2>&1 | grep -v amdgpu.ids | cut -c1-30,118-200 >> gpurun_out/wg2_ab.txt
done
done
(cd r-super_amd/csrc && rm -f _build/conv3d_wgrad2.o && make > /dev/null 2>&1)
