cd $GRAFT_REPO_ROOT
export RSUPER_IGEMM_VARIANT=5 RSUPER_PC2_PROF=1
for L in "3 dgrad" "0 dgrad"; do echo "== $L"; timeout 300 python tools/prof_one.py $L 2>&1 | grep pc2_prof | tail -8 | sed -n '1p;5p'; done
