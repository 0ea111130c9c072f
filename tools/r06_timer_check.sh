#!/bin/bash
# The roofline pass's timing events: unfenced (hipEventDisableSystemFence, default) vs torch's default events, per-step conv times, and agreement with rocprofv3.
for f in 0 0 0 1 1; do RSUPER_TIMER_FENCED=$f python bench.py --no-cpu-baseline --no-secondary --steps 20 --warmup 6 2>/dev/null | python -c "
import json,sys;d=json.loads(sys.stdin.read().strip().splitlines()[-1]);r=d['roofline'];print('fenced=$f', round(d['ms_per_step'],3), 'frac', round(r['frac'],4), 'median-step conv', round(r['conv_ms_per_step'],3), 'mean', round(r['conv_ms_per_step_mean'],3), 'each', r['conv_ms_each_step'], {k:(round(v['avg_us'],1), round(v['max_us'],0)) for k,v in r['per_kernel'].items()})"; done
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats -d $GRAFT_REPO_ROOT/gpurun_out/prof_tc -o tc -- python $GRAFT_REPO_ROOT/bench.py --no-cpu-baseline --no-secondary --steps 20 --warmup 6 > $GRAFT_REPO_ROOT/gpurun_out/tc_bench.json 2>/dev/null
cd $GRAFT_REPO_ROOT
python -c "
import json;d=json.loads(open('gpurun_out/tc_bench.json').read().strip().splitlines()[-1]);r=d['roofline'];print('under rocprofv3:', round(d['ms_per_step'],3), 'frac', round(r['frac'],4), 'conv', round(r['conv_ms_per_step'],3), {k:round(v['avg_us'],1) for k,v in r['per_kernel'].items()})"
ls gpurun_out/prof_tc | head; f=$(ls gpurun_out/prof_tc/*kernel_stats.csv | head -1); python - "$f" <<'P'
import csv,sys
rows=list(csv.DictReader(open(sys.argv[1])))
conv=[r for r in rows if any(k in r['Name'] for k in ('igemm','wgrad'))]
tot=sum(float(r['TotalDurationNs']) for r in conv); n=sum(int(r['Calls']) for r in conv)
steps=36
print('rocprofv3 kernel stats: conv-family kernels', round(tot/1e6/steps,3), 'ms/step over', steps, 'steps (', n/steps, 'launches/step )')
for r in sorted(conv,key=lambda r:-float(r['TotalDurationNs']))[:12]: print(' ', r['Name'][:70], r['Calls'], round(float(r['AverageNs'])/1e3,1))
P
