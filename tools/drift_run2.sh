set -x
bash tools/drift_ensemble.sh 2>&1 | tail -40
T=/tmp/g; mkdir -p $T
python tools/drift.py --mode f32 --steps 1 --grads $T/f32.pt > /dev/null 2>&1
python tools/drift.py --mode bf16 --steps 1 --grads $T/head.pt > /dev/null 2>&1
RSUPER_KD=0 python tools/drift.py --mode bf16 --steps 1 --grads $T/kd0.pt > /dev/null 2>&1
RSUPER_IGEMM_VARIANT=0 python tools/drift.py --mode bf16 --steps 1 --grads $T/v0.pt > /dev/null 2>&1
python tools/drift.py --root .bis/86d1453 --mode bf16 --steps 1 --grads $T/r04.pt > /dev/null 2>&1
python tools/drift.py --root .bis/86d1453 --mode f32 --steps 1 --grads $T/r04f32.pt > /dev/null 2>&1
python tools/grad_compare.py $T/f32.pt head=$T/head.pt r04=$T/r04.pt kd0=$T/kd0.pt v0=$T/v0.pt r04f32=$T/r04f32.pt 2>&1 | tee gpurun_out/grad_compare.txt
