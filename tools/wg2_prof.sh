# cycle stamps of conv3d_wgrad2.hip's tile loop (block 0, every wave): loop body vs barrier wait per tile; builds with -DWG2_PROF [+ extra flags] on the box
cd $GRAFT_REPO_ROOT
: > gpurun_out/wg2_prof.txt
for FL in "" "$@"; do
  (cd r-super_amd/csrc && rm -f _build/conv3d_wgrad2.o && make WG2_EXTRA="-DWG2_PROF $FL" > /dev/null 2>&1)
  echo "#### [-DWG2_PROF $FL]" >> gpurun_out/wg2_prof.txt
  RSUPER_WG2_PROF=1 BC_ONLY=${BC_ONLY:-"inc,up4.0"} python tools/bench_conv.py bf16 2>&1 | grep "wg2_prof" | awk '!seen[$0]++' | head -6 >> gpurun_out/wg2_prof.txt
done
(cd r-super_amd/csrc && rm -f _build/conv3d_wgrad2.o && make > /dev/null 2>&1)
