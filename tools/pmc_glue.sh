# HBM-side traffic of the attention-stage kernels (separate --pmc passes, kernel-trace only) -> tools/pmc_glue_summary.py
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d gpurun_out/pmc_glue_fetch -o r -- python tools/pmc_glue.py > /dev/null 2>&1
rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d gpurun_out/pmc_glue_write -o r -- python tools/pmc_glue.py > /dev/null 2>&1
python tools/pmc_glue_summary.py gpurun_out/pmc_glue_fetch gpurun_out/pmc_glue_write > gpurun_out/pmc_glue.md
rm -rf gpurun_out/pmc_glue_fetch gpurun_out/pmc_glue_write
