"""A/B of flash_fwd64_kernel's ring depth in one process: NSLOT = 2 (shipped: one barrier per 64-key tile) vs
NSLOT = 4 (VITA_ATTN64_RING=4: four-slot K / V rings, a barrier every TWO tiles), interleaved rounds on random data, outputs compared
bit for bit.  Prints JSON lines; writes gpurun_out/r06_attn_ring.jsonl."""
import json, os, sys
os.environ.setdefault("VITA_DEBUG", "1")
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from long_vita_amd import ops
DEV = "cuda"
os.makedirs(os.path.join(ROOT, "gpurun_out"), exist_ok=True)
LOG = open(os.path.join(ROOT, "gpurun_out", "r06_attn_ring.jsonl"), "a")
Hq, Hkv, D = 40, 8, 128
# generalised: --var NAME A B switches any developer variable between two values (default: the ring depth 2 vs 4)
VAR, VA, VB = "VITA_ATTN64_RING", "2", "4"
argv = sys.argv[1:]
if argv[:1] == ["--var"]:
    VAR, VA, VB = argv[1:4]
    argv = argv[4:]
for S in [int(x) for x in (argv or ["16384", "32768", "131072"])]:
    g = torch.Generator(device=DEV).manual_seed(S)
    q = torch.randn(1, S, Hq, D, generator=g, device=DEV).bfloat16()
    k = torch.randn(1, S, Hkv, D, generator=g, device=DEV).bfloat16()
    v = torch.randn(1, S, Hkv, D, generator=g, device=DEV).bfloat16()
    outs, times = {}, {VA: [], VB: []}
    for rnd in range(4):
        for ring in (VA, VB):
            os.environ[VAR] = ring
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
    rec = {"kind": "attn64_ab", "var": VAR, "S": S, "ms_" + VA: sorted(times[VA]), "ms_" + VB: sorted(times[VB]),
           "tflops_%s_median" % VA: fl / sorted(times[VA])[1] / 1e9, "tflops_%s_median" % VB: fl / sorted(times[VB])[1] / 1e9,
           "bit_identical": bool(torch.equal(outs[VA], outs[VB])),
           "max_abs_diff": float((outs[VA].float() - outs[VB].float()).abs().max())}
    print(json.dumps(rec), flush=True)
    LOG.write(json.dumps(rec) + "\n"); LOG.flush()
