"""Aggregate rocprofv3 counter_collection.csv files: mean counter value per kernel per dispatch."""
import csv, glob, sys, collections
def main(root):
    agg = collections.defaultdict(lambda: collections.defaultdict(list))
    for f in glob.glob(root + "/**/*counter_collection.csv", recursive=True):
        for row in csv.DictReader(open(f)):
            k = row["Kernel_Name"]
            if "attn_bwd_kvp" in k: k = "attn_bwd_kvp (dK + dV)"
            elif "attn_bwd_dq64" in k: k = "attn_bwd_dq64 (dQ)"
            elif "attn_delta" in k: k = "attn_delta (pre-pass)"
            elif "attn_bwd" in k: k = k.split("::")[-1].split("(")[0][:60]
            elif "flash_fwd64v" in k: k = "flash_fwd64v (d = 64, non-causal)"
            elif "flash_fwd64" in k: k = "flash_fwd64"
            elif "flash_fwd" in k: k = "flash_fwd<" + ("128" if "128" in k else "64") + ">"
            elif "gemm_bf16" in k: k = "gemm_bf16<" + k.split("<")[1].split(">")[0] + ">"
            elif "gemm_w4" in k: k = "gemm_w4<" + k.split("<")[1].split(">")[0] + ">"
            elif "Cijk" in k: k = "vendor " + k[:48]
            else: continue
            agg[k][row["Counter_Name"]].append(float(row["Counter_Value"]))
    for k in sorted(agg):
        print(k)
        for c in sorted(agg[k]):
            v = agg[k][c]
            print(f"   {c:28s} n={len(v):3d} mean={sum(v)/len(v):.6g}")
main(sys.argv[1])
