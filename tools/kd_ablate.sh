# Ablation builds of the depth-reuse igemm on the GPU box: where the time of an item goes (usage: bash tools/kd_ablate.sh [layer filter] ["flags" ...])
cd $GRAFT_REPO_ROOT/r-super_amd/csrc
L=${1:-up4}
shift
if [ $# -eq 0 ]; then set -- "" "-DKD_SKIP_STAGE" "-DKD_SKIP_EPI" "-DKD_SKIP_STAGE -DKD_SKIP_EPI" "-DKD_SKIP_MMA" "-DKD_SKIP_MMA -DKD_SKIP_EPI"; fi
for F in "$@"; do
  rm -f _build/conv3d_igemm_kd.o
  make KD_EXTRA="$F" > /dev/null 2>&1
  echo "== flags: $F"
  (cd ../.. && BC_ONLY=$L timeout 300 python tools/bench_conv.py 2>&1 | grep -v "^#" | grep -v amdgpu.ids)
done
rm -f _build/conv3d_igemm_kd.o; make > /dev/null 2>&1
