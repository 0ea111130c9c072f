#!/bin/bash
# MedFormer step: pointwise / depthwise checks, then eager step time with the single-slab pointwise weight gradient on / off
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out
timeout 600 python -m pytest tests -x -q -m gpu -k "pointwise or depthwise or medformer" 2>&1 | tail -3
for e in "" "RSUPER_PW_WG_DIRECT=0"; do
  echo "== $e"; env $e timeout 600 python tools/medformer_step.py 20 bf16 2>&1 | tail -1
  env $e timeout 600 python tools/medformer_step.py 20 bf16 graph 2>&1 | tail -1
done
