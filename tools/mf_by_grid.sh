cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
timeout 250 rocprofv3 --kernel-trace --output-format csv -d /tmp/kt_bg -o r -- python tools/medformer_step.py 8 bf16 > /dev/null 2>&1
python tools/kernel_by_grid.py $(ls /tmp/kt_bg/*kernel_trace.csv | head -1) "battn|depthwise_fwd|depthwise_wgrad_kernel|cnorm_stats" 12 2>&1 | head -40
