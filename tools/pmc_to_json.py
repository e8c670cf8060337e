"""profiles/<tag>_attn128k_pmc_raw.txt (tools/collect_profiles.sh: three SEPARATE rocprofv3 --pmc passes summarised by tools/pmc_summary.py)
+ the kernel-trace average of flash_fwd64_kernel -> profiles/<tag>_attn128k_pmc.json, the file bench.py reads `roofline.traffic` from.

Corrections (MI355X_MICROARCH.md, HBM / rocprofv3): FETCH_SIZE and WRITE_SIZE are in KB; on gfx950 FETCH_SIZE counts a 128-byte request as
64 bytes for wide coalesced reads, so the fetch figure is doubled; WRITE_SIZE is taken as is; Infinity-Cache hits count as fetches.
usage: python tools/pmc_to_json.py <raw.txt> <kernel_stats.txt> <out.json> [method note]"""
import json, re, sys

def parse(raw):
    out, cur = {}, None
    for line in open(raw):
        if not line.startswith(" "):
            cur = line.strip(); out.setdefault(cur, {})
        else:
            m = re.match(r"\s+(\S+)\s+n=\s*(\d+)\s+mean=(\S+)", line)
            if m and cur:
                out[cur][m.group(1)] = float(m.group(3))
    return out

def kernel_ms(stats, needle):
    """tools/rocpd_summary.py line: name, calls, total ms, AVERAGE ms, min, max, percent."""
    for line in open(stats):
        if needle in line:
            f = line.split()
            try:
                return float(f[-4])
            except (ValueError, IndexError):
                continue
    return None

def main():
    raw, stats, dst = sys.argv[1:4]
    note = sys.argv[4] if len(sys.argv) > 4 else ""
    c = parse(raw)
    att = c.get("flash_fwd64", {})
    S, H, D = 131072, 40, 128
    fetch = att["FETCH_SIZE"] * 1024 * 2
    write = att["WRITE_SIZE"] * 1024
    ms = kernel_ms(stats, "flash_fwd64")
    doc = {"kernel": "flash_fwd64_kernel (d = 128, causal; 4 waves x 64 rows)", "seq": S, "n_gpus": 1, "heads": "40:8", "head_dim": D,
           "FETCH_SIZE_KB": att["FETCH_SIZE"], "WRITE_SIZE_KB": att["WRITE_SIZE"], "fetch_bytes_corrected": fetch, "write_bytes": write,
           "hbm_bytes_per_launch": fetch + write,
           "algorithmic_bytes_per_launch": 2 * S * H * D * 2 + 2 * S * 8 * D * 2,
           "method": "rocprofv3 --pmc FETCH_SIZE, --pmc WRITE_SIZE and the SQ / GRBM set in three SEPARATE passes (tools/collect_profiles.sh -> "
                     "tools/pmc_kernels.py, PMC_S=131072), mean of 3 launches; FETCH_SIZE doubled per MI355X_MICROARCH.md HBM section (gfx950 counts "
                     "128-B requests at 64 B for wide coalesced reads); WRITE_SIZE uncalibrated; Infinity-Cache hits are counted as fetches. " + note}
    sq = {k: v for k, v in att.items() if k not in ("FETCH_SIZE", "WRITE_SIZE")}
    if sq and ms:
        clk = sq["GRBM_GUI_ACTIVE"] / 8 / (ms * 1e-3) / 1e9          # the counter is summed over the 8 XCDs
        sq.update(ms_per_launch_rocprofv3_kernel_trace=ms, effective_clock_ghz=clk,
                  mfma_busy_frac="SQ_VALU_MFMA_BUSY_CYCLES / (128 GRBM_GUI_ACTIVE)",
                  mfma_busy_frac_value=sq["SQ_VALU_MFMA_BUSY_CYCLES"] / (128 * sq["GRBM_GUI_ACTIVE"]))
        sq["valu_insts_per_mfma"] = sq["SQ_INSTS_VALU"] / sq["SQ_INSTS_MFMA"]
        sq["wait_any_frac"] = sq["SQ_WAIT_ANY"] / sq["SQ_WAVE_CYCLES"]
        sq["wait_inst_frac"] = sq["SQ_WAIT_INST_ANY"] / sq["SQ_WAVE_CYCLES"]
        sq["active_inst_frac"] = sq["SQ_ACTIVE_INST_ANY"] / sq["SQ_WAVE_CYCLES"]
        sq["mfma_flops_executed"] = sq["SQ_INSTS_MFMA"] * 32768
        doc["sq_counters_S131072"] = sq
    for name, v in c.items():
        if name.startswith("gemm_w4") and "FETCH_SIZE" in v:
            g = dict(v)
            g["hbm_bytes_per_launch"] = v["FETCH_SIZE"] * 1024 * 2 + v["WRITE_SIZE"] * 1024
            g["algorithmic_bytes_per_launch"] = (131072 * 5120 + 2 * 13824 * 5120 + 131072 * 13824) * 2
            doc["gemm_w4_fc1_swiglu_M131072"] = g
    json.dump(doc, open(dst, "w"), indent=1)
    print(json.dumps({k: doc[k] for k in ("hbm_bytes_per_launch", "algorithmic_bytes_per_launch")}))

main()
