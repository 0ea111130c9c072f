cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
rocprofv3 --kernel-trace --output-format csv -d /tmp/kt_mf -o r -- python tools/medformer_step.py 8 bf16 > gpurun_out/mf_eager.log 2>&1
python tools/kernel_stats.py $(ls /tmp/kt_mf/*kernel_trace.csv | head -1) 12 > gpurun_out/kstats_mf.txt
tail -1 gpurun_out/mf_eager.log
