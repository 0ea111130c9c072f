# same-box test of a build flag of the second-generation weight gradient: bash tools/wg2_prio.sh "flags" ...
cd $GRAFT_REPO_ROOT/r-super_amd/csrc
for F in "$@"; do
  rm -f _build/conv3d_wgrad2.o; make WG2_EXTRA="$F" > /dev/null 2>&1
  echo "== flags: $F"
  (cd ../.. && BC_ONLY=inc,up4,up3,up2 timeout 300 python tools/bench_conv.py 2>&1 | grep -v "^#" | grep -v amdgpu.ids)
done
rm -f _build/conv3d_wgrad2.o; make > /dev/null 2>&1
