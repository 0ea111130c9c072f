# Cycle stamps of the depth-reuse igemm (KD_PROF build) on the GPU box: per wave and item -- MFMA loop, epilogue, barrier wait
# usage: bash tools/kd_prof.sh [layer] ["extra flags" ...]   (one build + run per flag set)
cd $GRAFT_REPO_ROOT/r-super_amd/csrc
L=${1:-up4}; shift
if [ $# -eq 0 ]; then set -- ""; fi
for F in "$@"; do
  rm -f _build/conv3d_igemm_kd.o; make KD_EXTRA="-DKD_PROF $F" > /dev/null 2>&1
  echo "== flags: $F"
  (cd ../.. && RSUPER_KD_PROF=1 BC_ONLY=$L timeout 300 python tools/bench_conv.py 2>&1 | grep -v "^#" | grep -v amdgpu.ids | grep -v "^$" | tail -14 | cut -c1-400)
done
rm -f _build/conv3d_igemm_kd.o; make > /dev/null 2>&1
