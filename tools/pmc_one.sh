# usage: bash tools/pmc_one.sh "<layer> <fwd|dgrad|wgrad>" ...   (SQ counters of one conv kernel, separate --pmc passes)
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
for L in "$@"; do
  set -- $L
  rocprofv3 --pmc SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_VALU_MFMA_BUSY_CYCLES --kernel-trace --output-format csv -d gpurun_out/pc_${1}_${2}${3}_a -o r -- python tools/prof_one.py $1 $2 $3 > /dev/null 2>&1
  rocprofv3 --pmc SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS SQ_INSTS_VMEM_RD SQ_INSTS_SALU SQ_INSTS_VMEM_WR --kernel-trace --output-format csv -d gpurun_out/pc_${1}_${2}${3}_b -o r -- python tools/prof_one.py $1 $2 $3 > /dev/null 2>&1
  rocprofv3 --pmc SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_MISC SQ_INST_LEVEL_VMEM SQ_INST_LEVEL_LDS --kernel-trace --output-format csv -d gpurun_out/pc_${1}_${2}${3}_c -o r -- python tools/prof_one.py $1 $2 $3 > /dev/null 2>&1
  rocprofv3 --pmc FETCH_SIZE GRBM_GUI_ACTIVE --kernel-trace --output-format csv -d gpurun_out/pc_${1}_${2}${3}_d -o r -- python tools/prof_one.py $1 $2 $3 > /dev/null 2>&1
  rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d gpurun_out/pc_${1}_${2}${3}_e -o r -- python tools/prof_one.py $1 $2 $3 > /dev/null 2>&1
done
