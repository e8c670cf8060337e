#!/bin/bash
# Back-pressure counters (r06): does a memory instruction that cannot issue (TA / LDS FIFO full) explain the GEMM loop's issue stalls?
# fc2-shaped NT GEMM with the library given in VITA_HIP_LIB (default: the shipped one), and the 128K attention forward.
set -u
OUT=$PWD/gpurun_out; mkdir -p $OUT; export TMPDIR=/tmp; R=$PWD
F=$OUT/${1:-r06}_fifo_pmc.txt
echo "==== library: ${VITA_HIP_LIB:-shipped} ====" | tee -a $F
for C in "SQ_VMEM_TA_ADDR_FIFO_FULL SQ_VMEM_TA_CMD_FIFO_FULL SQ_LDS_CMD_FIFO_FULL SQ_LDS_DATA_FIFO_FULL SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_MFMA GRBM_GUI_ACTIVE" "SQ_VALU_MFMA_COEXEC_CYCLES SQ_INST_CYCLES_VMEM_RD SQ_INST_LEVEL_VMEM SQ_INST_LEVEL_LDS SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_LDS SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_INST_LDS"; do
  D=/tmp/prof_fifo; rm -rf $D
  (cd /tmp && PMC_SHAPE=0 timeout 300 rocprofv3 --pmc $C --output-format csv -d $D -- python $R/tools/pmc_gemm_vs_vendor.py > /dev/null 2> $D.err) || tail -3 $D.err
  python tools/pmc_summary.py $D | tee -a $F
  rm -rf $D
  (cd /tmp && PMC_S=131072 PMC_GEMM=0 timeout 300 rocprofv3 --pmc $C --output-format csv -d $D -- python $R/tools/pmc_kernels.py > /dev/null 2> $D.err) || tail -3 $D.err
  python tools/pmc_summary.py $D | tee -a $F
done
