# same-box test of a build flag of the producer/consumer igemm: bash tools/pc_prio.sh "flags" ...
cd $GRAFT_REPO_ROOT/r-super_amd/csrc
for F in "$@"; do
  rm -f _build/conv3d_igemm.o; make CL_EXTRA="$F" > /dev/null 2>&1
  echo "== flags: $F"
  (cd ../.. && RSUPER_SPLIT_DGRAD=0 BC_ONLY=up4,down1,64,up3 timeout 300 python tools/bench_conv.py 2>&1 | grep -v "^#" | grep -v amdgpu.ids | cut -c1-40,75-118)
done
rm -f _build/conv3d_igemm.o; make > /dev/null 2>&1
