#!/bin/bash
# first invocation on a fresh box of the selection that failed; $1 = extra environment (A=B)
cd "$GRAFT_REPO_ROOT" || exit 1
env $1 timeout 300 python -m pytest tests -x -q -m gpu -k "norm or pointwise or depthwise or medformer" 2>&1 | grep -v "Warning\|warnings.warn" | grep -E "^E  |passed|failed" | cut -c1-900 | head -12
