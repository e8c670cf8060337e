"""gpurun_out/<tag>_attn_bwd16k_pmc_raw.txt (tools/collect_bwd_pmc.sh) -> profiles/<tag>_attn_bwd16k_pmc.txt: derived figures per kernel + the raw means.
usage: python tools/pmc_bwd_summary.py RAW OUT ["note"]"""
import re, sys

raw, out = sys.argv[1], sys.argv[2]
note = sys.argv[3] if len(sys.argv) > 3 else ""
data, cur = {}, None
for line in open(raw):
    m = re.match(r"\s+(\w+)\s+n=\s*\d+\s+mean=([\d.e+-]+)", line)
    if m:
        data[cur][m.group(1)] = float(m.group(2))
    elif line.strip():
        cur = line.strip(); data.setdefault(cur, {})
rows = []
for k, c in data.items():
    mf = c.get("SQ_INSTS_MFMA", 0.0)
    rows.append({"kernel": k, "lds_bank_conflict_cycles": c.get("SQ_LDS_BANK_CONFLICT"), "mfma_instructions": mf,
                 "valu_per_mfma": round(c["SQ_INSTS_VALU"] / mf, 3) if mf else None,
                 "mfma_busy": round(c["SQ_VALU_MFMA_BUSY_CYCLES"] / (128 * c["GRBM_GUI_ACTIVE"]), 4),
                 "gpu_cycles_per_xcd": c["GRBM_GUI_ACTIVE"] / 8, "wait_any_frac": round(c["SQ_WAIT_ANY"] / c["SQ_WAVE_CYCLES"], 4),
                 "fetch_gb": round(2 * c["FETCH_SIZE"] * 1024 / 1e9, 4), "write_gb": round(c["WRITE_SIZE"] * 1024 / 1e9, 4)})
with open(out, "w") as f:
    f.write("# attention backward at S = 16384, 40 : 8 heads, d = 128: rocprofv3 --pmc passes (tools/collect_bwd_pmc.sh; four separate runs), mean per launch\n"
            "# derived: mfma_busy = SQ_VALU_MFMA_BUSY_CYCLES / (128 GRBM_GUI_ACTIVE); fetch bytes = 2 x FETCH_SIZE KB (gfx950 correction)\n")
    if note:
        f.write("# " + note + "\n")
    f.write("\n")
    for r in rows:
        f.write(repr(r) + "\n")
    f.write("\n# raw\n" + open(raw).read())
for r in rows:
    print(r)
