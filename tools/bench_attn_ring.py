"""A/B of flash_fwd64_kernel's ring depth in one process: NSLOT = 2 (shipped: one barrier per 64-key tile) vs
NSLOT = 4 (VITA_ATTN64_RING=4: four-slot K / V rings, a barrier every TWO tiles), interleaved rounds on random data, outputs compared
bit for bit.  Prints JSON lines; writes gpurun_out/r03_attn_ring.jsonl."""
import json, os, sys
os.environ.setdefault("VITA_DEBUG", "1")
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from long_vita_amd import ops
DEV = "cuda"
os.makedirs(os.path.join(ROOT, "gpurun_out"), exist_ok=True)
LOG = open(os.path.join(ROOT, "gpurun_out", "r03_attn_ring.jsonl"), "a")
Hq, Hkv, D = 40, 8, 128
for S in [int(x) for x in (sys.argv[1:] or ["16384", "32768", "131072"])]:
    g = torch.Generator(device=DEV).manual_seed(S)
    q = torch.randn(1, S, Hq, D, generator=g, device=DEV).bfloat16()
    k = torch.randn(1, S, Hkv, D, generator=g, device=DEV).bfloat16()
    v = torch.randn(1, S, Hkv, D, generator=g, device=DEV).bfloat16()
    outs, times = {}, {"2": [], "4": []}
    for rnd in range(4):
        for ring in ("2", "4"):
            os.environ["VITA_ATTN64_RING"] = ring
            o = torch.empty_like(q)
            ops.flash_attn(q, k, v, causal=True, out=o)          # warm
            a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            a.record()
            for _ in range(2):
                ops.flash_attn(q, k, v, causal=True, out=o)
            b.record(); torch.cuda.synchronize()
            times[ring].append(a.elapsed_time(b) / 2)
            outs[ring] = o
    fl = 4.0 * D * Hq * S * (S + 1) / 2
    rec = {"kind": "attn64_ring_ab", "S": S, "ms_ring2": sorted(times["2"]), "ms_ring4": sorted(times["4"]),
           "tflops_ring2_median": fl / sorted(times["2"])[1] / 1e9, "tflops_ring4_median": fl / sorted(times["4"])[1] / 1e9,
           "bit_identical": bool(torch.equal(outs["2"], outs["4"])),
           "max_abs_diff": float((outs["2"].float() - outs["4"].float()).abs().max())}
    print(json.dumps(rec), flush=True)
    LOG.write(json.dumps(rec) + "\n"); LOG.flush()
