#!/bin/bash
# bench.py --no-pool: same-box A/B of one environment switch, two repetitions.  Usage: bash tools/nopool_ab2.sh VAR valueA valueB
cd "$GRAFT_REPO_ROOT" || exit 1
for rep in 1 2; do
  for val in $2 $3; do
    env $1=$val python bench.py --no-pool --no-cpu-baseline --no-secondary --steps 30 --warmup 5 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$1=$val', round(d['ms_per_step'],3), round(d['roofline']['conv_ms_per_step'],3), {k: round(v['avg_us'],1) for k,v in d['roofline']['per_kernel'].items()})"
  done
done
