# rocprofv3 PMC passes (separate runs, no trace domains mixed in) for single conv kernels; see tools/prof_one.py
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
for L in "0 fwd" "0 dgrad" "0 wgrad" "1 fwd"; do
  set -- $L
  rocprofv3 --pmc SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_VALU_MFMA_BUSY_CYCLES --kernel-trace --output-format csv -d gpurun_out/pmc_${1}_${2}_a -o r -- python tools/prof_one.py $1 $2 > /dev/null 2>&1
  rocprofv3 --pmc SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS SQ_INSTS_VMEM_RD SQ_INSTS_SALU SQ_INSTS_VMEM_WR --kernel-trace --output-format csv -d gpurun_out/pmc_${1}_${2}_b -o r -- python tools/prof_one.py $1 $2 > /dev/null 2>&1
  rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d gpurun_out/pmc_${1}_${2}_c -o r -- python tools/prof_one.py $1 $2 > /dev/null 2>&1
  rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d gpurun_out/pmc_${1}_${2}_d -o r -- python tools/prof_one.py $1 $2 > /dev/null 2>&1
done
