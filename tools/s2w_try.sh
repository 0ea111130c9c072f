#!/bin/bash
# strided weight gradient: parity checks + layer times (bf16, f32)
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out
timeout 300 python - > gpurun_out/s2w_checks.txt 2>&1 <<'PY'
import sys, os
sys.path.insert(0, 'tests'); sys.path.insert(0, '.')
import gpu_checks as g
for m in ('f32', 'bf16'):
    for a in [(m, 1, (8, 8, 32), 32, 32, 0), (m, 2, (12, 10, 20), 16, 32, 32), (m, 1, (7, 9, 35), 8, 16, 16), (m, 2, (5, 17, 66), 40, 24, 24),
              (m, 1, (2, 3, 5), 8, 8, 8), (m, 3, (24, 24, 24), 64, 128, 128)]:
        r = g.check_wgrad_s2(*a)
        print(r['name'], 'OK' if r['ok'] else 'FAIL', r['note'], flush=True)
PY
BC_ONLY_S2=1 timeout 300 python tools/bench_conv.py bf16 > gpurun_out/s2w_bf16.txt 2>&1
BC_ONLY_S2=1 timeout 300 python tools/bench_conv.py f32 > gpurun_out/s2w_f32.txt 2>&1
tail -n 20 gpurun_out/s2w_checks.txt gpurun_out/s2w_bf16.txt gpurun_out/s2w_f32.txt
