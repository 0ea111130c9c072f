# same-box test of a build flag of the weight-stationary igemm: bash tools/ws_prio.sh "flags" ...
cd $GRAFT_REPO_ROOT/r-super_amd/csrc
for F in "$@"; do
  rm -f _build/conv3d_igemm_ws.o; make WS_EXTRA="$F" > /dev/null 2>&1
  echo "== flags: $F"
  (cd ../.. && BC_ONLY=inc timeout 300 python tools/bench_conv.py 2>&1 | grep -v "^#" | grep -v amdgpu.ids)
done
rm -f _build/conv3d_igemm_ws.o; make > /dev/null 2>&1
