# Round-4 evidence in one GPU call; everything lands in gpurun_out/ (copy the summaries to profiles/):
#   per-layer conv table (default / round-3 weight gradient), kernel stats + launch sequence of the bench step, FETCH / WRITE traffic of the step,
#   SQ counters of the second-generation weight gradient, the 1-rank --force-ddp trace (gradient exchange vs backward), the default bench line.
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
timeout 300 python tools/bench_conv.py bf16 > gpurun_out/r04_conv_layers.txt 2>/dev/null
RSUPER_WGRAD2=0 BC_ONLY="->" timeout 300 python tools/bench_conv.py bf16 2>/dev/null | cut -c1-40,118-200 > gpurun_out/r04_conv_layers_wgrad_r03.txt
timeout 300 bash tools/kstats.sh r04
timeout 300 bash tools/kseq.sh r04
timeout 600 bash tools/pmc_step.sh
python tools/pmc_step_summary.py gpurun_out/pmc_step_fetch gpurun_out/pmc_step_write 6 r04 > gpurun_out/r04_pmc_step.md
cp profiles/conv_traffic.json gpurun_out/conv_traffic.json
rm -rf gpurun_out/pmc_step_fetch gpurun_out/pmc_step_write
timeout 600 bash tools/pmc_one.sh "0 wgrad" "1 wgrad" "4 wgrad" > /dev/null 2>&1
for x in 0_wgrad 1_wgrad 4_wgrad; do echo "## layer $x"; python tools/pmc_summary.py gpurun_out/pc_$x; done > gpurun_out/r04_pmc_wgrad2.md
rm -rf gpurun_out/pc_*
rm -rf /tmp/ddp_tr; timeout 300 rocprofv3 --kernel-trace --output-format csv -d /tmp/ddp_tr -o r -- python bench.py --force-ddp --steps 6 --warmup 3 --roofline-steps 0 --no-secondary --no-cpu-baseline > gpurun_out/r04_force_ddp.json 2>/dev/null
python tools/ddp_overlap.py $(ls /tmp/ddp_tr/*kernel_trace.csv | head -1) > gpurun_out/r04_ddp_overlap.txt 2>&1
timeout 1500 python bench.py > gpurun_out/r04_bench_default.json 2> gpurun_out/r04_bench_default.err
tail -c 300 gpurun_out/r04_bench_default.json
timeout 300 python tools/ddp_bucket_timeline.py 6 > gpurun_out/r04_ddp_bucket_timeline.txt 2>&1
timeout 300 python tools/xhat_probe.py > gpurun_out/r04_xhat_probe.txt 2>&1
timeout 300 bash tools/wg2_prof.sh "-DWG2_SKIP_STAGE" > /dev/null 2>&1; cp gpurun_out/wg2_prof.txt gpurun_out/r04_wg2_prof.txt
timeout 600 bash tools/mf_kstats.sh > /dev/null 2>&1
