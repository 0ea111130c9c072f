cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
timeout 200 rocprofv3 --kernel-trace --output-format csv -d /tmp/kt_lp -o r -- python tools/loss_phases.py 10 > /dev/null 2>&1
python tools/kernel_stats.py $(ls /tmp/kt_lp/*kernel_trace.csv | head -1) 24 > gpurun_out/kstats_loss.txt
head -30 gpurun_out/kstats_loss.txt | cut -c1-150
