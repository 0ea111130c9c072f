cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
# kernel statistics of the eager MedFormer step (rocprofv3 --kernel-trace); $1 = extra environment (A=B), $2 = output suffix
rm -rf /tmp/kt_mf
env $1 rocprofv3 --kernel-trace --output-format csv -d /tmp/kt_mf -o r -- python tools/medformer_step.py 8 bf16 > gpurun_out/mf_eager$2.log 2>&1
python tools/kernel_stats.py $(ls /tmp/kt_mf/*kernel_trace.csv | head -1) 12 > gpurun_out/kstats_mf$2.txt
tail -1 gpurun_out/mf_eager$2.log
