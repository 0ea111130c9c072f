#!/bin/bash
# MedFormer: the semantic-map product and the padded aux head on the pointwise kernels (round 6) against the library GEMMs they replace.
python -m pytest tests -m gpu -q -x -k "medformer or pointwise" 2>&1 | tail -4
for i in 1 2; do
for cfg in "RSUPER_MF_MAP_PRODUCT=1 RSUPER_MF_PAD_HEAD=1" "RSUPER_MF_MAP_PRODUCT=0 RSUPER_MF_PAD_HEAD=0" "RSUPER_MF_MAP_PRODUCT=1 RSUPER_MF_PAD_HEAD=0"; do
  echo "== $cfg"; env $cfg python tools/medformer_step.py 12 bf16 2>&1 | tail -2
  env $cfg python tools/medformer_step.py 14 bf16 graph 2>&1 | tail -1
done; done
