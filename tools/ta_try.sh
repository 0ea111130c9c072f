#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out
timeout 300 python - > gpurun_out/ta_checks.txt 2>&1 <<'PY'
import sys
sys.path.insert(0, 'tests'); sys.path.insert(0, '.')
import gpu_checks as g
for a in [(2, 81, 10, 32), (1, 128, 2, 32), (3, 7, 3, 16), (2, 65, 4, 24), (1, 1, 1, 4), (2, 96, 2, 64)]:
    r = g.check_token_attn(*a)
    print(r['name'], 'OK' if r['ok'] else 'FAIL', r['note'], flush=True)
PY
cat gpurun_out/ta_checks.txt
for e in 1 0; do
  RSUPER_MF_TOKEN_ATTN=$e timeout 600 python tools/medformer_step.py 12 bf16 2>&1 | tail -1
done
