cd $GRAFT_REPO_ROOT
for m in 0 auto 1; do echo "overlap=$m"; RSUPER_WGRAD_OVERLAP=$m timeout 600 python bench.py --no-cpu-baseline --no-secondary --steps 40 --warmup 8 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(round(d['ms_per_step'],3), round(d['roofline']['frac'],4), round(d['roofline']['conv_ms_per_step'],3))"; done
