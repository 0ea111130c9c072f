cd $GRAFT_REPO_ROOT
python - <<P 2>&1 | grep -v amdgpu
import sys; sys.path.insert(0,"tests"); sys.path.insert(0,"tests/golden")
import gpu_checks as gc
for m in ("f32","bf16"):
    for fn,a in [(gc.check_stem,(m,64,12)),(gc.check_head,(m,64,70,10)),(gc.check_head,(m,16,130,9)),(gc.check_head,(m,32,42,12)),(gc.check_stem,(m,32,16)),(gc.check_unet_wide,(m,))]:
        r=fn(*a); print(r["ok"], r["name"], r["err"], r["note"])
P
