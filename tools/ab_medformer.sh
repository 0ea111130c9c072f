# Same-box A/B of the MedFormer step: .ab/ (built copy of an earlier commit) vs the working tree, interleaved
for rep in 1 2; do
  for d in .ab .; do
    (cd $d && python tools/medformer_step.py 10 bf16 graph 2>/dev/null | tail -1 | sed "s|^|$d |")
  done
done
