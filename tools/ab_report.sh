# Same-box A/B of config 3 (report supervision on): .ab/ (built copy of an earlier commit) vs the working tree, interleaved
for rep in 1 2; do
  for d in .ab .; do
    (cd $d && python bench.py --report --no-cpu-baseline --no-secondary --steps 30 --warmup 5 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$d config3', round(d['ms_per_step'],3))")
  done
done
