#!/bin/bash
# MedFormer step: norm / pointwise / depthwise checks, then the replayed step time over RSUPER_CNORM_SMALL_VOX
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out
timeout 600 python -m pytest tests -x -q -m gpu -k "norm or pointwise or depthwise or medformer" 2>&1 | tail -3
for e in "RSUPER_CNORM_SMALL_VOX=512" "RSUPER_CNORM_SMALL_VOX=2048" "RSUPER_CNORM_SMALL_VOX=16384" "RSUPER_CNORM_SMALL_VOX=512"; do
  echo "== $e"; env $e timeout 600 python tools/medformer_step.py 20 bf16 graph 2>&1 | tail -1
done
