# Same-box kernel-level A/B: rocprofv3 kernel trace of .ab/ (built copy of an earlier commit) and of the working tree, then tools/kdiff.py
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
for d in .ab .; do
  (cd $d && rocprofv3 --kernel-trace --output-format csv -d /tmp/kt_$$ -o r -- python bench.py --steps 10 --warmup 3 --roofline-steps 0 --no-secondary --no-cpu-baseline > /dev/null 2>&1)
  n=$( [ $d = . ] && echo B || echo A )
  python tools/kernel_stats.py $(ls /tmp/kt_$$/*kernel_trace.csv | head -1) 13 > gpurun_out/kstats_$n.txt
  rm -rf /tmp/kt_$$
done
python tools/kdiff.py gpurun_out/kstats_A.txt gpurun_out/kstats_B.txt > gpurun_out/kdiff.txt
