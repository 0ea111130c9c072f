# same-box A/B of the per-layer conv times (forward / data gradient / weight gradient): .ab/ (built copy of an earlier commit) vs the working tree
# usage: bash tools/ab_conv.sh   (BC_ONLY selects layers)   -> gpurun_out/ab_conv.txt
cd $GRAFT_REPO_ROOT
: > gpurun_out/ab_conv.txt
for rep in 1 2; do
  for d in .ab .; do
    echo "#### [$d] run $rep" >> gpurun_out/ab_conv.txt
    (cd $d && BC_ONLY=${BC_ONLY:-"->"} python tools/bench_conv.py bf16 2>&1 | grep -v "amdgpu.ids\|stride-2" | cut -c1-22,38-200) >> gpurun_out/ab_conv.txt
  done
done
cat gpurun_out/ab_conv.txt
