#!/bin/bash
# bf16-vs-f32 loss drift of the headline workload on ONE box: HEAD with each dispatch switch, and built copies of earlier commits (.bis/<sha>).
# Usage (on the GPU box): bash tools/drift_bisect.sh [steps]   -> gpurun_out/drift.jsonl + gpurun_out/drift_summary.txt
STEPS=${1:-70}
OUT=gpurun_out/drift.jsonl
mkdir -p gpurun_out; : > $OUT
run() { # label, root, mode, env...
  local label=$1 root=$2 mode=$3; shift 3
  echo "== $label" >&2
  env "$@" RSUPER_LABEL="$label" timeout 600 python tools/drift.py --root $root --mode $mode --steps $STEPS --out $OUT > /dev/null 2>gpurun_out/drift_err_$$.txt || { echo "FAILED $label"; tail -3 gpurun_out/drift_err_$$.txt; }
}
run head_f32 . f32
run head_bf16 . bf16
run head_bf16_again . bf16
run head_kd0 . bf16 RSUPER_KD=0
run head_mt1_0 . bf16 RSUPER_WG2_MT1=0
run head_split0 . bf16 RSUPER_SPLIT_DGRAD=0
run head_wg2off . bf16 RSUPER_WGRAD2=0
run head_variant0 . bf16 RSUPER_IGEMM_VARIANT=0
run head_variant2 . bf16 RSUPER_IGEMM_VARIANT=2
run head_nobox . bf16 RSUPER_NO_BOX=1
for c in $(ls .bis 2>/dev/null); do
  run bis_${c}_bf16 .bis/$c bf16
done
run r04_f32 .bis/86d1453 f32
python - <<'EOF' | tee gpurun_out/drift_summary.txt
import json
recs = [json.loads(l) for l in open('gpurun_out/drift.jsonl')]
f32 = next(r for r in recs if r['env'].get('RSUPER_LABEL') == 'head_f32')['loss']
print(f"{'label':28s} {'L35':>8s} {'d35':>8s} {'L70':>8s} {'d70':>8s}  mean|d| over steps 20..70")
for r in recs:
    l = r['loss']; n = len(l)
    i35, i70 = min(34, n - 1), n - 1
    md = sum(l[i] - f32[i] for i in range(20, n)) / max(1, n - 20)
    print(f"{r['env'].get('RSUPER_LABEL', '?'):28s} {l[i35]:8.4f} {l[i35] - f32[i35]:+8.4f} {l[i70]:8.4f} {l[i70] - f32[i70]:+8.4f}  {md:+.4f}")
EOF
