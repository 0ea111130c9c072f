# Same-box A/B/C of an environment switch: bash tools/ab_env3.sh VAR v1 v2 v3 ...
V=$1; shift
for rep in 1 2; do
  for val in "$@"; do
    env $V=$val python bench.py --no-cpu-baseline --no-secondary --steps 30 --warmup 5 --roofline-steps 0 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$V=$val', round(d['ms_per_step'],3))"
  done
done
