cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
# kernel statistics + idle gaps of the config-3 step (report supervision)
rm -rf /tmp/kt_c3
timeout 300 rocprofv3 --kernel-trace --output-format csv -d /tmp/kt_c3 -o r -- python bench.py --report --steps 12 --warmup 5 --no-secondary --no-cpu-baseline --roofline-steps 0 > gpurun_out/c3_bench.log 2>&1
python tools/kernel_stats.py $(ls /tmp/kt_c3/*kernel_trace.csv | head -1) 17 > gpurun_out/kstats_c3.txt
grep -n "idle between" -A22 gpurun_out/kstats_c3.txt | cut -c1-170
