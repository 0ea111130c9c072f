cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
timeout 200 rocprofv3 --kernel-trace --output-format csv -d /tmp/kt_ov -o r -- python tools/mode_consistency.py unet 96 26 30 seg step > /dev/null 2>&1
python tools/graph_overlap.py $(ls /tmp/kt_ov/*kernel_trace.csv | head -1) 2>&1 | head -40
