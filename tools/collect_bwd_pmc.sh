#!/bin/bash
# PMC passes of the attention backward kernels at S = 16384 (40 : 8 heads, d = 128): separate rocprofv3 --pmc runs (never combined with tracing),
# summarised per kernel by tools/pmc_summary.py.   usage: tools/collect_bwd_pmc.sh <tag>  -> gpurun_out/<tag>_attn_bwd16k_pmc_raw.txt
set -u
TAG=${1:-r05}
OUT=$PWD/gpurun_out
mkdir -p $OUT
export TMPDIR=/tmp
R=$PWD
rm -f $OUT/${TAG}_attn_bwd16k_pmc_raw.txt
for C in "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_LDS SQ_ACTIVE_INST_LDS" "SQ_VALU_MFMA_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_MFMA GRBM_GUI_ACTIVE" FETCH_SIZE WRITE_SIZE; do
  D=/tmp/prof_bwd_$(echo $C | cut -d' ' -f1); rm -rf $D
  (cd /tmp && PMC_S=16384 PMC_GEMM=0 PMC_BWD=1 timeout 200 rocprofv3 --pmc $C --output-format csv -d $D -- python $R/tools/pmc_kernels.py > /dev/null 2> $D.err) || tail -3 $D.err
  python tools/pmc_summary.py $D | tee -a $OUT/${TAG}_attn_bwd16k_pmc_raw.txt
done
