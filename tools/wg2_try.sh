# second-generation weight gradient: parity checks + per-layer times (new kernel / RSUPER_WGRAD2=0 = the round-3 kernel)
cd $GRAFT_REPO_ROOT
python tests/gpu_diag.py check_conv_bwd > gpurun_out/wg2_diag.txt 2>&1
python tests/gpu_diag.py check_wgrad_xhat >> gpurun_out/wg2_diag.txt 2>&1
grep -c PASS gpurun_out/wg2_diag.txt > gpurun_out/wg2_summary.txt; grep "FAIL\|ERROR\|SUMMARY" gpurun_out/wg2_diag.txt | cut -c1-220 >> gpurun_out/wg2_summary.txt
BC_ONLY=${BC_ONLY:-"->"} python tools/bench_conv.py bf16 2>&1 | grep -v amdgpu.ids | cut -c1-40,118-200 > gpurun_out/wg2_layers.txt
RSUPER_WGRAD2=0 BC_ONLY=${BC_ONLY:-"->"} python tools/bench_conv.py bf16 2>&1 | grep -v amdgpu.ids | cut -c1-40,118-200 > gpurun_out/wg2_layers_old.txt
