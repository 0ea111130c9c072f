# Round-3 closing evidence (no PMC passes: the conv kernels did not change since profiles/r03_pmc_*): per-layer conv table incl. the strided
# rows, kernel stats of the UNet and the MedFormer step, the default bench line.  Everything lands in gpurun_out/; copy the summaries to profiles/.
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
timeout 200 python tools/bench_conv.py bf16 > gpurun_out/r03_conv_layers_box.txt 2>/dev/null
RSUPER_NO_BOX=1 timeout 200 python tools/bench_conv.py bf16 2>/dev/null | grep -v "s2\|stride-2" > gpurun_out/r03_conv_layers_nobox.txt
timeout 200 python tools/bench_conv.py f32 2>/dev/null | grep "s2\|stride-2" > gpurun_out/r03_conv_layers_s2_f32.txt
timeout 300 bash tools/kstats.sh r03
timeout 300 bash tools/mf_kstats.sh
timeout 600 python bench.py > gpurun_out/r03_bench_default.json 2> gpurun_out/r03_bench_default.err
tail -c 400 gpurun_out/r03_bench_default.json
