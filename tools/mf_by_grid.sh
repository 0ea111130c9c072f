cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
# per launch shape (grid size) of selected kernels in the eager MedFormer step; $1 = kernel name regex
rm -rf /tmp/kt_bg
timeout 250 rocprofv3 --kernel-trace --output-format csv -d /tmp/kt_bg -o r -- python tools/medformer_step.py 8 bf16 > /dev/null 2>&1
python tools/kernel_by_grid.py $(ls /tmp/kt_bg/*kernel_trace.csv | head -1) "${1:-battn|depthwise_fwd|depthwise_wgrad_kernel|cnorm_stats}" 12 2>&1 | head -${2:-40}
