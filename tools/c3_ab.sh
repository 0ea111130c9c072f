#!/bin/bash
# config 3 (report supervision): loss tests, then bench.py --report with the supervision prefetch on / off (same box, twice each)
cd "$GRAFT_REPO_ROOT" || exit 1
[ -n "$SKIP_TESTS" ] || timeout 600 python -m pytest tests -x -q -m gpu -k "loss or report or ball or isolate or golden or train_step" 2>&1 | grep -E "passed|failed|^E  " | head -5
for e in 1 0 1 0 1 0; do
  RSUPER_PREFETCH_SUPERVISION=$e timeout 300 python bench.py --report --steps 60 --warmup 10 --no-secondary --no-cpu-baseline 2>/dev/null | python -c "
import sys, json
for l in sys.stdin:
    if l.startswith('{'):
        j = json.loads(l); print('prefetch $e: ms/step %.3f  final_loss %s' % (j['ms_per_step'], j['config']['final_loss']))
"
done
