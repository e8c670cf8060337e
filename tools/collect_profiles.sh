#!/bin/bash
# Round evidence on the GPU box (run through gpurun from the repo root): bench line, rocprofv3 kernel stats of the same command,
# PMC passes (FETCH_SIZE / WRITE_SIZE / SQ counters in SEPARATE runs, never combined with tracing) of the dominant kernel.
# usage: tools/collect_profiles.sh <tag>    -> gpurun_out/<tag>_*
set -u
TAG=${1:-r02}
OUT=gpurun_out
mkdir -p $OUT
export TMPDIR=/tmp
python bench.py --steps 3 --warmup 1 > $OUT/${TAG}_bench128k_n1.json 2> $OUT/${TAG}_bench128k_n1.err
tail -c 3000 $OUT/${TAG}_bench128k_n1.json
rm -rf /tmp/prof_ks; rocprofv3 --kernel-trace --stats -d /tmp/prof_ks -o ks -- python bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-parity-check > $OUT/${TAG}_bench128k_under_rocprof.json 2> /tmp/prof_ks.err
DB=$(find /tmp/prof_ks -name "*.db" | head -1)
if [ -n "$DB" ]; then python tools/rocpd_summary.py $DB $OUT/${TAG}_bench128k_kernel_stats.txt | head -12; else find /tmp/prof_ks | head; CSV=$(find /tmp/prof_ks -name "*kernel_stats.csv" | head -1); [ -n "$CSV" ] && cp $CSV $OUT/${TAG}_bench128k_kernel_stats.csv && head -8 $CSV; fi
for C in FETCH_SIZE WRITE_SIZE "SQ_VALU_MFMA_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_MFMA GRBM_GUI_ACTIVE"; do
  D=/tmp/prof_pmc_$(echo $C | cut -d' ' -f1); rm -rf $D
  PMC_S=131072 PMC_GEMM=1 rocprofv3 --pmc $C --output-format csv -d $D -- python tools/pmc_kernels.py > /dev/null 2> $D.err
  python tools/pmc_summary.py $D | tee -a $OUT/${TAG}_attn128k_pmc_raw.txt
done
