# Kernel-level stats of the bench step in the working tree: rocprofv3 kernel trace -> gpurun_out/kstats_<tag>.txt   (usage: bash tools/kstats.sh <tag>)
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
rocprofv3 --kernel-trace --output-format csv -d /tmp/kt_$$ -o r -- python bench.py --steps 10 --warmup 3 --roofline-steps 0 --no-secondary --no-cpu-baseline > /dev/null 2>&1
python tools/kernel_stats.py $(ls /tmp/kt_$$/*kernel_trace.csv | head -1) 13 > gpurun_out/kstats_$1.txt
rm -rf /tmp/kt_$$
