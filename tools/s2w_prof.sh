#!/bin/bash
# kernel durations of the strided rows of tools/bench_conv.py (rocprofv3 --kernel-trace)
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
mkdir -p gpurun_out
BC_ONLY_S2=1 timeout 300 rocprofv3 --kernel-trace --output-format csv -d /tmp/s2wprof -o s2w -- python tools/bench_conv.py ${1:-bf16} > gpurun_out/s2w_prof_run.txt 2>&1
python tools/kernel_stats.py $(find /tmp/s2wprof -name '*kernel_trace.csv' | head -1) 1 > gpurun_out/s2w_prof_stats.txt
head -40 gpurun_out/s2w_prof_stats.txt
