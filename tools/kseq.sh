# launch sequence of one bench step (default workload) -> gpurun_out/kseq_<tag>.txt   (usage: bash tools/kseq.sh <tag> [bench args])
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
TAG=$1; shift
rocprofv3 --kernel-trace --output-format csv -d /tmp/ks_$$ -o r -- python bench.py --steps 6 --warmup 3 --roofline-steps 0 --no-secondary --no-cpu-baseline "$@" > /dev/null 2>&1
python tools/kernel_seq.py $(ls /tmp/ks_$$/*kernel_trace.csv | head -1) 9 > gpurun_out/kseq_$TAG.txt
rm -rf /tmp/ks_$$
