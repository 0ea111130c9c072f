# Cycle stamps of the weight-gradient kernel's tile loop (block 0): rebuilds conv3d_wgrad.o with -DRS_WG_PROF on the box, runs one layer shape per call
# usage: bash tools/wg_prof.sh   -> gpurun_out/wg_prof.txt
cd $GRAFT_REPO_ROOT/r-super_amd/csrc && rm -f _build/conv3d_wgrad.o && make WG_EXTRA="-DRS_WG_PROF $WG_MORE" > /dev/null 2>&1; cd $GRAFT_REPO_ROOT
RSUPER_WG_PROF=1 python tools/bench_conv.py bf16 2>&1 | grep -v "s2 " > gpurun_out/wg_prof.txt
