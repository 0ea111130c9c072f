# usage: bash tools/kt_one.sh "<layer> <fwd|dgrad|wgrad> [xhat]" ...   -> gpurun_out/kt_one.txt: per-kernel average durations (rocprofv3 kernel trace) of each configuration
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
: > gpurun_out/kt_one.txt
for L in "$@"; do
  rm -rf /tmp/kt1; PROF_ITERS=10 rocprofv3 --kernel-trace --output-format csv -d /tmp/kt1 -o r -- python tools/prof_one.py $L > /dev/null 2>&1
  echo "== $L" >> gpurun_out/kt_one.txt
  python tools/kernel_stats.py $(ls /tmp/kt1/*kernel_trace.csv | head -1) 10 2>/dev/null | grep -v "^#\|^$\|^after\|elementwise\|idle" | head -8 | cut -c1-150 >> gpurun_out/kt_one.txt
done
