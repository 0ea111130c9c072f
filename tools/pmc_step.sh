# HBM-side traffic of one training step (separate --pmc passes, kernel-trace only) -> tools/pmc_step_summary.py
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d gpurun_out/pmc_step_fetch -o r -- python bench.py --steps 4 --warmup 2 --roofline-steps 0 --no-secondary --no-cpu-baseline > /dev/null 2>&1
rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d gpurun_out/pmc_step_write -o r -- python bench.py --steps 4 --warmup 2 --roofline-steps 0 --no-secondary --no-cpu-baseline > /dev/null 2>&1
