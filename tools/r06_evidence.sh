# Round-6 evidence in one GPU call: per-layer conv table, kernel statistics + launch sequence of the bench step, HBM-side traffic of the step (separate
# --pmc passes, kernel-trace only), SQ / memory counters of every forward and data-gradient igemm family, the default bench line.
# Everything lands in gpurun_out/r06_*; the summaries are copied to profiles/ by hand.
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
rm -f gpurun_out/r06_pmc_igemm.md
timeout 300 python tools/bench_conv.py bf16 > gpurun_out/r06_conv_layers.txt 2>/dev/null
rocprofv3 --kernel-trace --output-format csv -d /tmp/kt_r06 -o r -- python bench.py --steps 10 --warmup 3 --roofline-steps 0 --no-secondary --no-cpu-baseline > /dev/null 2>&1
python tools/kernel_stats.py $(ls /tmp/kt_r06/*kernel_trace.csv | head -1) 13 > gpurun_out/r06_bench_kernel_stats.txt
python tools/kernel_seq.py $(ls /tmp/kt_r06/*kernel_trace.csv | head -1) 13 > gpurun_out/r06_bench_kernel_seq.txt 2>/dev/null
rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/kt_r06s -o r -- python bench.py --steps 10 --warmup 3 --roofline-steps 0 --no-secondary --no-cpu-baseline > /dev/null 2>&1
cp $(ls /tmp/kt_r06s/*kernel_stats.csv | head -1) gpurun_out/r06_bench_rocprofv3_kernel_stats.csv 2>/dev/null
rm -rf gpurun_out/pmc_step_fetch gpurun_out/pmc_step_write
timeout 600 bash tools/pmc_step.sh
python tools/pmc_step_summary.py gpurun_out/pmc_step_fetch gpurun_out/pmc_step_write 6 r06 > gpurun_out/r06_pmc_step.md 2>&1; cp profiles/conv_traffic.json gpurun_out/conv_traffic.json
for L in "0 fwd" "0 dgrad" "1 fwd" "1 dgrad" "3 fwd" "3 dgrad" "4 fwd" "4 dgrad" "5 fwd" "5 dgrad"; do
  set -- $L
  rm -rf gpurun_out/pmc5_${1}_${2}_*
  rocprofv3 --pmc SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_VALU_MFMA_BUSY_CYCLES --kernel-trace --output-format csv -d gpurun_out/pmc5_${1}_${2}_a -o r -- python tools/prof_one.py $1 $2 > /dev/null 2>&1
  rocprofv3 --pmc SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS SQ_INSTS_VMEM_RD SQ_INSTS_SALU SQ_INSTS_VMEM_WR SQ_INSTS_MFMA --kernel-trace --output-format csv -d gpurun_out/pmc5_${1}_${2}_b -o r -- python tools/prof_one.py $1 $2 > /dev/null 2>&1
  rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d gpurun_out/pmc5_${1}_${2}_c -o r -- python tools/prof_one.py $1 $2 > /dev/null 2>&1
  rocprofv3 --pmc WRITE_SIZE GRBM_GUI_ACTIVE --kernel-trace --output-format csv -d gpurun_out/pmc5_${1}_${2}_d -o r -- python tools/prof_one.py $1 $2 > /dev/null 2>&1
  echo "## layer $1 $2" >> gpurun_out/r06_pmc_igemm.md
  python tools/pmc_summary.py gpurun_out/pmc5_${1}_${2} >> gpurun_out/r06_pmc_igemm.md 2>&1
  rm -rf gpurun_out/pmc5_${1}_${2}_*
done
for L in "fwd 96 32 64" "dgrad 96 32 64" "fwd 48 64 128" "dgrad 48 64 128"; do
  set -- $L
  rm -rf gpurun_out/pmc5s_${1}_${2}_*
  rocprofv3 --pmc SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_VALU_MFMA_BUSY_CYCLES --kernel-trace --output-format csv -d gpurun_out/pmc5s_${1}_${2}_a -o r -- python tools/prof_s2.py $1 $2 $3 $4 > /dev/null 2>&1
  rocprofv3 --pmc SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS SQ_INSTS_VMEM_RD SQ_INSTS_SALU SQ_INSTS_VMEM_WR SQ_INSTS_MFMA --kernel-trace --output-format csv -d gpurun_out/pmc5s_${1}_${2}_b -o r -- python tools/prof_s2.py $1 $2 $3 $4 > /dev/null 2>&1
  rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d gpurun_out/pmc5s_${1}_${2}_c -o r -- python tools/prof_s2.py $1 $2 $3 $4 > /dev/null 2>&1
  rocprofv3 --pmc WRITE_SIZE GRBM_GUI_ACTIVE --kernel-trace --output-format csv -d gpurun_out/pmc5s_${1}_${2}_d -o r -- python tools/prof_s2.py $1 $2 $3 $4 > /dev/null 2>&1
  echo "## strided [conv1 | shortcut] $1, input $2^3, $3 -> 2 x $4" >> gpurun_out/r06_pmc_igemm.md
  python tools/pmc_summary.py gpurun_out/pmc5s_${1}_${2} >> gpurun_out/r06_pmc_igemm.md 2>&1
  rm -rf gpurun_out/pmc5s_${1}_${2}_*
done
rm -rf gpurun_out/pmc_step_fetch gpurun_out/pmc_step_write
timeout 900 python bench.py > gpurun_out/r06_bench_default.json 2> gpurun_out/r06_bench_default.err
tail -c 600 gpurun_out/r06_bench_default.json

# MedFormer (4 warm-up + 12 timed eager steps = 16 in the trace): kernel statistics of the eager step (the network R-Super trains)
rocprofv3 --kernel-trace --output-format csv -d /tmp/kt_r06m -o r -- python tools/medformer_step.py 12 bf16 > /dev/null 2>&1
python tools/kernel_stats.py $(ls /tmp/kt_r06m/*kernel_trace.csv | head -1) 16 > gpurun_out/r06_medformer_kernel_stats.txt 2>/dev/null
# per-launch rates of the InstanceNorm-backward tail (bytes = 3 or 4 tensors of the layer x 2 B) from the bench trace
python tools/in_bwd_rates.py > gpurun_out/r06_in_bwd_rates.txt 2>/dev/null
python -c "import __graft_entry__ as g; g.smoke()" > gpurun_out/r06_smoke.txt 2>&1
python -m pytest tests -m gpu -q -rs > gpurun_out/r06_gputest_full.txt 2>&1
grep -v "Warning\|^  \|^$\|warnings.warn" gpurun_out/r06_gputest_full.txt | tail -25 > gpurun_out/r06_gputest.txt
tail -3 gpurun_out/r06_gputest.txt
