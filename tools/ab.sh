# A/B on one box: .ab/ (built copy of an earlier commit) vs the working tree, interleaved twice.
for rep in 1 2; do
  for d in .ab .; do
    (cd $d && python bench.py --no-cpu-baseline --steps 30 --warmup 5 | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$d', round(d['ms_per_step'],3), {k: round(v['avg_us'],1) for k,v in d['roofline']['per_kernel'].items()})")
  done
done
