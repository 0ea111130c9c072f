# Where does the producer/consumer weight-gradient kernel (RSUPER_WGRAD_CFG0=0, 32-row layers) spend its time?  full / consumers alone / producers alone
cd $GRAFT_REPO_ROOT
run() { (cd r-super_amd/csrc && rm -f _build/conv3d_wgrad.o && make WG_EXTRA="$1" > /dev/null 2>&1); echo "== build [$1] CFG0=$2"; BC_ONLY=inc RSUPER_WGRAD_CFG0=$2 python tools/bench_conv.py bf16 2>&1 | grep "^inc" | cut -c1-40,118-200; }
run "" 2 > gpurun_out/wg_pc_ablate.txt
run "" 1 >> gpurun_out/wg_pc_ablate.txt
run "" 0 >> gpurun_out/wg_pc_ablate.txt
run "-DRS_WG_SKIP_PROD" 0 >> gpurun_out/wg_pc_ablate.txt
run "-DRS_WG_SKIP_CONS" 0 >> gpurun_out/wg_pc_ablate.txt
