cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
for L in "1 wgrad"; do
  set -- $L
  rocprofv3 --pmc SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_VALU_MFMA_BUSY_CYCLES --kernel-trace --output-format csv -d gpurun_out/pc_${1}_${2}_a -o r -- python tests/prof_one.py $1 $2 > /dev/null 2>&1
  rocprofv3 --pmc SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS SQ_INSTS_VMEM_RD SQ_INSTS_SALU SQ_INSTS_VMEM_WR --kernel-trace --output-format csv -d gpurun_out/pc_${1}_${2}_b -o r -- python tests/prof_one.py $1 $2 > /dev/null 2>&1
done
