"""Developer aid: per-phase shader-clock breakdown of flash_fwd_kernel (VITA_ATTN_VARIANT bit 3)."""
import ctypes as C, os, sys
os.environ.setdefault("VITA_DEBUG", "1")      # developer switches (VITA_GEMM_*, VITA_ATTN_*) are honoured only with this set
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from long_vita_amd import ops, lib
S = int(os.environ.get("PMC_S", "32768"))
q = torch.randn(1, S, 40, 128, device="cuda").bfloat16(); k = torch.randn(1, S, 8, 128, device="cuda").bfloat16()
v = torch.randn(1, S, 8, 128, device="cuda").bfloat16(); o = torch.empty_like(q)
h = lib.load()
fn = h.vita_debug_attn_timing
fn.argtypes = [C.c_void_p, C.c_int]
ops.flash_attn(q, k, v, causal=True, out=o); torch.cuda.synchronize()
buf = (C.c_ulonglong * 16)()
fn(None, 1)
ops.flash_attn(q, k, v, causal=True, out=o); torch.cuda.synchronize()
fn(buf, 0)
for g, name in ((0, "group A (waves 0-3)"), (8, "group B (waves 4-7)")):
    top, qk, sm, bar, n = [buf[g + i] for i in range(5)]
    tot = top + qk + sm + bar
    if n:
        print(f"{name}: tiles={n} cyc/tile={tot/n:.0f}  top={top/n:.0f} qk={qk/n:.0f} sm_pv={sm/n:.0f} barrier={bar/n:.0f}")
