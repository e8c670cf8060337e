#!/bin/bash
# PMC passes of the NT GEMM on fc2 / qkv@16K shapes, library kernel vs the vendor's (tools/pmc_gemm_vs_vendor.py); per kernel AND grid size.
set -u
TAG=${1:-r05}
OUT=$PWD/gpurun_out; mkdir -p $OUT; export TMPDIR=/tmp; R=$PWD
rm -f $OUT/${TAG}_gemm_pmc_raw.txt
for SH in 0 1; do
echo "==== shape $SH: $([ $SH = 0 ] && echo 'fc2 at 128K: M 131072, N 5120, K 13824' || echo 'qkv at 16K: M 16384, N 7168, K 5120') ====" | tee -a $OUT/${TAG}_gemm_pmc_raw.txt
for C in "SQ_VALU_MFMA_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_MFMA GRBM_GUI_ACTIVE" "SQ_BUSY_CYCLES SQ_WAVES SQ_INSTS_SALU SQ_INSTS_LDS SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT" FETCH_SIZE; do
  D=/tmp/prof_gemm_$(echo $C | cut -d' ' -f1); rm -rf $D
  (cd /tmp && PMC_SHAPE=$SH timeout 300 rocprofv3 --pmc $C --output-format csv -d $D -- python $R/tools/pmc_gemm_vs_vendor.py > /dev/null 2> $D.err) || tail -3 $D.err
  python - $D <<'PY' | tee -a $OUT/${TAG}_gemm_pmc_raw.txt
import csv, glob, sys, collections
agg = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob(sys.argv[1] + "/**/*counter_collection.csv", recursive=True):
    for row in csv.DictReader(open(f)):
        k = row["Kernel_Name"]
        if "gemm_w4" in k or "gemm_bf16" in k: name = "ours " + k.split("(")[0][-40:]
        elif "Cijk" in k: name = "vendor " + k[:60]
        else: continue
        agg[(name, row.get("Grid_Size", "?"), row.get("Workgroup_Size", "?"))][row["Counter_Name"]].append(float(row["Counter_Value"]))
for k in sorted(agg):
    print(k)
    for c in sorted(agg[k]):
        v = agg[k][c]
        print(f"   {c:28s} n={len(v):3d} mean={sum(v)/len(v):.6g}")
PY
done
done
