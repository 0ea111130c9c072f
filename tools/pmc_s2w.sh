# SQ / memory counters of the strided weight-gradient kernel (conv3d_wgrad_s2.hip): separate --pmc passes, kernel trace only, every pass under its own timeout
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
run() {  # tag, command...
  tag=$1; shift
  timeout 120 rocprofv3 --pmc SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_VALU_MFMA_BUSY_CYCLES --kernel-trace --output-format csv -d gpurun_out/pn_${tag}_a -o r -- "$@" > /dev/null 2>&1
  timeout 120 rocprofv3 --pmc SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS SQ_INSTS_VMEM_RD SQ_INSTS_SALU SQ_INSTS_VMEM_WR --kernel-trace --output-format csv -d gpurun_out/pn_${tag}_b -o r -- "$@" > /dev/null 2>&1
  timeout 120 rocprofv3 --pmc FETCH_SIZE GRBM_GUI_ACTIVE --kernel-trace --output-format csv -d gpurun_out/pn_${tag}_d -o r -- "$@" > /dev/null 2>&1
  timeout 120 rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d gpurun_out/pn_${tag}_e -o r -- "$@" > /dev/null 2>&1
  echo "## $tag"; python tools/pmc_summary.py gpurun_out/pn_$tag
}
{
run s2_wgrad_96 python tools/prof_s2.py wgrad 96 32 64
run s2_wgrad_48 python tools/prof_s2.py wgrad 48 64 128
} > gpurun_out/r03_pmc_s2_wgrad.md 2>&1
cat gpurun_out/r03_pmc_s2_wgrad.md | head -60
