#!/bin/bash
# Is the bf16-vs-f32 loss gap of ONE trajectory a property of the kernels or of the trajectory?  Ensemble over seeds (initial weights + batch) and over
# last-bit perturbations of the initial weights (1e-6 relative: below bf16 resolution), for HEAD (default dispatch), HEAD without the depth-reuse igemm,
# HEAD on the classic kernels only, and a built copy of the round-4 tree (.bis/86d1453).
# Usage (GPU box): bash tools/drift_ensemble.sh  -> gpurun_out/drift_ens.jsonl, gpurun_out/drift_ens_summary.txt
OUT=gpurun_out/drift_ens.jsonl
mkdir -p gpurun_out; : > $OUT
SEEDS=${SEEDS:-0,1,2,3,4,5}
PERT=${PERT:-0,1,2,3,4,5}
run() { # label root mode "extra args" [ENV=...]
  local label=$1 root=$2 mode=$3 extra=$4; shift 4; echo "== $label" >&2
  env "$@" RSUPER_LABEL="$label" timeout 1200 python tools/drift.py --root $root --mode $mode --steps 70 --out $OUT $extra > /dev/null 2>gpurun_out/drift_err.txt \
    || { echo "FAILED $label"; tail -3 gpurun_out/drift_err.txt; }
}
R04=.bis/86d1453
run f32        . f32  "--seeds $SEEDS"
run head       . bf16 "--seeds $SEEDS"
run kd0        . bf16 "--seeds $SEEDS" RSUPER_KD=0
run variant0   . bf16 "--seeds $SEEDS" RSUPER_IGEMM_VARIANT=0
[ -d $R04 ] && run r04 $R04 bf16 "--seeds $SEEDS"
run f32_pert   . f32  "--perturb $PERT"
run head_pert  . bf16 "--perturb $PERT"
run kd0_pert   . bf16 "--perturb $PERT" RSUPER_KD=0
[ -d $R04 ] && run r04_pert $R04 bf16 "--perturb $PERT"
python - <<'EOF' | tee gpurun_out/drift_ens_summary.txt
import json, statistics as st
recs = [json.loads(l) for l in open('gpurun_out/drift_ens.jsonl')]
lab = lambda r: r['env'].get('RSUPER_LABEL', '?')
def table(f32lab, labs, key):
    f32 = {r[key]: r['loss'] for r in recs if lab(r) == f32lab}
    print(f"{'label':12s} {'n':>2s} | bf16 - f32 at step 35: mean   sd    min    max | at step 70: mean   sd    min    max | per run (35 / 70)")
    for L in labs:
        rs = [r for r in recs if lab(r) == L and r[key] in f32]
        if not rs:
            continue
        d35 = [r['loss'][34] - f32[r[key]][34] for r in rs]; d70 = [r['loss'][69] - f32[r[key]][69] for r in rs]
        sd = lambda v: st.pstdev(v) if len(v) > 1 else 0.0
        print(f"{L:12s} {len(rs):2d} | {st.mean(d35):+.4f} {sd(d35):.4f} {min(d35):+.4f} {max(d35):+.4f} | {st.mean(d70):+.4f} {sd(d70):.4f} {min(d70):+.4f} {max(d70):+.4f} | "
              + ' '.join(f'{a:+.3f}/{b:+.3f}' for a, b in zip(d35, d70)))
print('# (a) six seeds (initial weights + batch); every bf16 run against the f32 run of the same seed')
table('f32', ['head', 'kd0', 'variant0', 'r04'], 'seed')
print('# (b) seed 0, initial weights perturbed by 1e-6 relative (perturbation id 0 = unperturbed); against the f32 run with the SAME perturbation')
table('f32_pert', ['head_pert', 'kd0_pert', 'r04_pert'], 'pert')
f = [r for r in recs if lab(r) == 'f32_pert']
if f:
    base = next(r for r in f if r['pert'] == 0)['loss']
    print('# f32 runs against the unperturbed f32 run (how far 1e-6 moves the f32 trajectory itself): step 35 / 70')
    print('  ' + ' '.join(f"{r['loss'][34] - base[34]:+.4f}/{r['loss'][69] - base[69]:+.4f}" for r in f if r['pert']))
EOF
