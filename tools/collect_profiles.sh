#!/bin/bash
# Round evidence on the GPU box (run through gpurun from the repo root): bench line, rocprofv3 kernel stats of the same command,
# PMC passes (FETCH_SIZE / WRITE_SIZE / SQ counters in SEPARATE runs, never combined with tracing) of the dominant kernel, the
# attention / GEMM microbenchmarks of the round, the 16K line, and a roctx-marked dry run.
# usage: tools/collect_profiles.sh <tag>    -> gpurun_out/<tag>_*
set -u
TAG=${1:-r05}
OUT=gpurun_out
mkdir -p $OUT
export TMPDIR=/tmp
python bench.py --steps 3 --warmup 1 > $OUT/${TAG}_bench128k_n1.json 2> $OUT/${TAG}_bench128k_n1.err
tail -c 3000 $OUT/${TAG}_bench128k_n1.json
rm -rf /tmp/prof_ks; (cd /tmp && rocprofv3 --kernel-trace --stats -d /tmp/prof_ks -o ks -- python $OLDPWD/bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-parity-check > $OLDPWD/$OUT/${TAG}_bench128k_under_rocprof.json 2> /tmp/prof_ks.err)
DB=$(find /tmp/prof_ks -name "*.db" | head -1)
if [ -n "$DB" ]; then python tools/rocpd_summary.py $DB $OUT/${TAG}_bench128k_kernel_stats.txt | head -12; else find /tmp/prof_ks | head; CSV=$(find /tmp/prof_ks -name "*kernel_stats.csv" | head -1); [ -n "$CSV" ] && cp $CSV $OUT/${TAG}_bench128k_kernel_stats.csv && head -8 $CSV; fi
rm -f $OUT/${TAG}_attn128k_pmc_raw.txt
for C in FETCH_SIZE WRITE_SIZE "SQ_VALU_MFMA_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_MFMA GRBM_GUI_ACTIVE"; do
  D=/tmp/prof_pmc_$(echo $C | cut -d' ' -f1); rm -rf $D
  (cd /tmp && PMC_S=131072 PMC_GEMM=1 rocprofv3 --pmc $C --output-format csv -d $D -- python $OLDPWD/tools/pmc_kernels.py > /dev/null 2> $D.err)
  python tools/pmc_summary.py $D | tee -a $OUT/${TAG}_attn128k_pmc_raw.txt
done
python tools/pmc_to_json.py $OUT/${TAG}_attn128k_pmc_raw.txt $OUT/${TAG}_bench128k_kernel_stats.txt $OUT/${TAG}_attn128k_pmc.json "Collected with the round's final kernels."
# microbenchmarks: attention forward / backward (7-unit form and, for the A/B, the r03 three-launch form), GEMMs with the vendor beside them
rm -f $OUT/microbench.jsonl
python tools/microbench.py attn attn_bwd > /dev/null 2>&1; cp $OUT/microbench.jsonl $OUT/${TAG}_attn_bwd_microbench.jsonl
VITA_ATTN_BWD_KVP=0 python tools/microbench.py attn_bwd > /dev/null 2>&1; grep attn_bwd $OUT/microbench.jsonl | tail -6 | sed 's/"kind": "attn_bwd"/"kind": "attn_bwd", "VITA_ATTN_BWD_KVP": 0/' >> $OUT/${TAG}_attn_bwd_microbench.jsonl
rm -f $OUT/microbench.jsonl
python tools/microbench.py gemm peaks > /dev/null 2>&1; cp $OUT/microbench.jsonl $OUT/${TAG}_gemm.jsonl
python bench.py --seq 16384 --steps 5 --warmup 2 --no-cpu-baseline > $OUT/${TAG}_bench16k_n1.json 2> /dev/null; tail -c 600 $OUT/${TAG}_bench16k_n1.json
# roctx ranges (VITA_DEBUG): marker + kernel trace of a dry run
rm -rf /tmp/prof_mk; (cd /tmp && VITA_DEBUG=1 rocprofv3 --marker-trace --kernel-trace --stats -d /tmp/prof_mk -o mk --output-format csv -- python $OLDPWD/bench.py --dry-run --steps 1 --warmup 0 --no-parity-check > /dev/null 2> /tmp/prof_mk.err)
MK=$(find /tmp/prof_mk -name "*marker_api_trace.csv" | head -1); [ -n "$MK" ] && (head -1 $MK; grep -c . $MK; cut -d, -f1-4 $MK | sed -n 2,12p) > $OUT/${TAG}_roctx_ranges_dry_run.txt; cat $OUT/${TAG}_roctx_ranges_dry_run.txt 2>/dev/null | head -8; [ -z "$MK" ] && (find /tmp/prof_mk | head; tail -5 /tmp/prof_mk.err)
