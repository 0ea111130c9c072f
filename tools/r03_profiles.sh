# Round-3 evidence in one GPU call: per-layer conv table (default vs box kernel off), kernel stats of the bench step, PMC traffic of the step,
# PMC counters of the box kernel, the default bench line.  Everything lands in gpurun_out/; copy the summaries to profiles/.
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
python tools/bench_conv.py bf16 > gpurun_out/r03_conv_layers_box.txt 2>/dev/null
RSUPER_NO_BOX=1 python tools/bench_conv.py bf16 > gpurun_out/r03_conv_layers_nobox.txt 2>/dev/null
bash tools/kstats.sh r03
bash tools/pmc_step.sh
python tools/pmc_step_summary.py gpurun_out/pmc_step_fetch gpurun_out/pmc_step_write 6 r03 > gpurun_out/r03_pmc_step.md
cp profiles/conv_traffic.json gpurun_out/conv_traffic.json
rm -rf gpurun_out/pmc_step_fetch gpurun_out/pmc_step_write
bash tools/pmc_one.sh "5 fwd" "5 dgrad" "6 fwd" "7 fwd" "8 fwd" > /dev/null 2>&1
for x in 5_fwd 5_dgrad 6_fwd 7_fwd 8_fwd; do echo "## layer $x"; python tools/pmc_summary.py gpurun_out/pc_$x; done > gpurun_out/r03_pmc_conv_box.md
rm -rf gpurun_out/pc_*
python bench.py > gpurun_out/r03_bench_default.json 2> gpurun_out/r03_bench_default.err
tail -c 600 gpurun_out/r03_bench_default.json
