# same-box test of a build flag of the tile-streaming weight gradient: bash tools/wg_prio.sh "flags" ...
cd $GRAFT_REPO_ROOT/r-super_amd/csrc
for F in "$@"; do
  rm -f _build/conv3d_wgrad.o; make WG_EXTRA="$F" > /dev/null 2>&1
  echo "== flags: $F"
  (cd ../.. && BC_ONLY=down1,64,128 timeout 300 python tools/bench_conv.py 2>&1 | grep -v "^#" | grep -v amdgpu.ids | cut -c1-40,118-160)
done
rm -f _build/conv3d_wgrad.o; make > /dev/null 2>&1
