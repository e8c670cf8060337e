#!/bin/bash
# PMC passes of the ViT attention forward (253 frames x 1025 tokens, 16 x 64, non-causal): separate rocprofv3 --pmc runs.
# usage: tools/collect_vit_pmc.sh <tag> [VITA_ATTN64V=0 for the r01 kernel]  -> gpurun_out/<tag>_vit_attn_pmc_raw.txt
set -u
TAG=${1:-r05}
OUT=$PWD/gpurun_out
mkdir -p $OUT
export TMPDIR=/tmp
R=$PWD
rm -f $OUT/${TAG}_vit_attn_pmc_raw.txt
for C in "SQ_VALU_MFMA_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_MFMA GRBM_GUI_ACTIVE" "SQ_BUSY_CYCLES SQ_WAVES SQ_INSTS_SALU SQ_INSTS_LDS SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT" FETCH_SIZE; do
  D=/tmp/prof_vit_$(echo $C | cut -d' ' -f1); rm -rf $D
  (cd /tmp && PMC_VIT=1 timeout 200 rocprofv3 --pmc $C --output-format csv -d $D -- python $R/tools/pmc_kernels.py > /dev/null 2> $D.err) || tail -3 $D.err
  python tools/pmc_summary.py $D | tee -a $OUT/${TAG}_vit_attn_pmc_raw.txt
done
