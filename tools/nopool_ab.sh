#!/bin/bash
# bench.py --no-pool (variant: strided down blocks): persistent strided kernels (round 5) vs the parity-class kernels (RSUPER_S2K=0 / RSUPER_S2D=0) vs the full-resolution evaluation (RSUPER_S2_KERNEL=0)
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out; out=gpurun_out/nopool_ab.txt; : > $out
for e in "" "RSUPER_S2K=0 RSUPER_S2D=0" "RSUPER_S2K=0" "RSUPER_S2D=0" "RSUPER_S2_KERNEL=0"; do
  echo "== bench.py --no-pool --steps 20 --warmup 5 --no-secondary --no-cpu-baseline   $e" >> $out
  env $e timeout 600 python bench.py --no-pool --steps 20 --warmup 5 --no-secondary --no-cpu-baseline 2>/dev/null | python -c "
import sys, json
for l in sys.stdin:
    if l.startswith('{'):
        j = json.loads(l); print('ms/step %.2f  value %.3e voxels/s  conv %.2f ms  frac %.3f  final_loss %s' % (j['ms_per_step'], j['value'], j['roofline'].get('conv_ms_per_step', float('nan')), j['roofline']['frac'], j['config']['final_loss']))
" >> $out
done
cat $out
