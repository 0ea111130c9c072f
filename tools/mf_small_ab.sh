#!/bin/bash
# MedFormer step after a small-kernel change: the checks that cover it, then the replayed and the eager step time
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out
timeout 600 python -m pytest tests -x -q -m gpu -k "${1:-norm or pointwise or depthwise or medformer or squeeze or checks}" 2>&1 | tail -2
timeout 600 python tools/medformer_step.py 20 bf16 graph 2>&1 | tail -1
timeout 600 python tools/medformer_step.py 20 bf16 2>&1 | tail -1
