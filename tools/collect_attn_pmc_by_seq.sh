#!/bin/bash
# SQ / GRBM counters of flash_fwd64_kernel at several sequence lengths (r06: where does 16K lose against 32K / 128K?).  One pass, no tracing.
set -u
OUT=$PWD/gpurun_out; mkdir -p $OUT; export TMPDIR=/tmp; R=$PWD
F=$OUT/${1:-r06}_attn_pmc_by_seq.txt; : > $F
for S in 4096 16384 32768 131072; do
  echo "==== S = $S ====" | tee -a $F
  for C in "SQ_VALU_MFMA_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_MFMA GRBM_GUI_ACTIVE" "SQ_BUSY_CYCLES SQ_WAVES SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAIT_INST_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_ACTIVE_INST_VALU"; do
    D=/tmp/prof_attnseq_$S; rm -rf $D
    (cd /tmp && PMC_S=$S PMC_GEMM=0 timeout 300 rocprofv3 --pmc $C --output-format csv -d $D -- python $R/tools/pmc_kernels.py > /dev/null 2> $D.err) || tail -3 $D.err
    python tools/pmc_summary.py $D | tee -a $F
  done
done
