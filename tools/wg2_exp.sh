# usage: bash tools/wg2_exp.sh "<flags build 1>" "<flags build 2>" ... ; layers from WG2_LAYERS (default: "1 wgrad" "1 wgrad xhat" "0 wgrad" "0 wgrad xhat")
cd $GRAFT_REPO_ROOT
: > gpurun_out/wg2_exp.txt
for FL in "$@"; do
  (cd r-super_amd/csrc && rm -f _build/conv3d_wgrad2.o && make WG2_EXTRA="$FL" > /dev/null 2>&1)
  echo "#### build [$FL]" >> gpurun_out/wg2_exp.txt
  bash tools/kt_one.sh "1 wgrad" "1 wgrad xhat" "0 wgrad" "0 wgrad xhat" "3 wgrad"; grep "==\|wgrad2" gpurun_out/kt_one.txt | grep -v reduce | cut -c1-45,100-140 >> gpurun_out/wg2_exp.txt
done
