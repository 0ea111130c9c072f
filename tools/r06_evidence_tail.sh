#!/bin/bash
# the two evidence pieces that depend on each other: HBM-side conv traffic of the step (-> profiles/conv_traffic.json, read by bench.py) and the default bench line
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
rm -rf gpurun_out/pmc_step_fetch gpurun_out/pmc_step_write
timeout 600 bash tools/pmc_step.sh
python tools/pmc_step_summary.py gpurun_out/pmc_step_fetch gpurun_out/pmc_step_write 6 r06 > gpurun_out/r06_pmc_step.md 2>&1; cp profiles/conv_traffic.json gpurun_out/conv_traffic.json
rm -rf gpurun_out/pmc_step_fetch gpurun_out/pmc_step_write
timeout 1500 python bench.py > gpurun_out/r06_bench_default.json 2> gpurun_out/r06_bench_default.err
tail -c 400 gpurun_out/r06_bench_default.json
