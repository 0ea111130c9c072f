# same-box A/B of the per-layer weight-gradient times: .ab/ (built copy of an earlier commit) vs the working tree, interleaved twice
# usage: bash tools/ab_layers.sh   (BC_ONLY selects layers; default all stride-1 layers)   -> gpurun_out/ab_layers.txt
cd $GRAFT_REPO_ROOT
: > gpurun_out/ab_layers.txt
for rep in 1 2; do
  for d in .ab .; do
    echo "#### [$d] run $rep" >> gpurun_out/ab_layers.txt
    (cd $d && BC_ONLY=${BC_ONLY:-"->"} python tools/bench_conv.py bf16 2>&1 | grep -v "amdgpu.ids\|stride-2" | cut -c1-22,118-200) >> gpurun_out/ab_layers.txt
  done
done
