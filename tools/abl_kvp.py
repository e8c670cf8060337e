"""Developer timing of the dK + dV pass alone at 16K (VITA_HIP_LIB selects the build: timing-only ablations of attn_bwd_kvp.hip)."""
import os, sys, time
os.environ.setdefault("VITA_DEBUG", "1")
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT)
import torch
from long_vita_amd import ops
S, Hq, Hkv, D = 16384, 40, 8, 128
q = torch.randn(1, S, Hq, D, device="cuda").bfloat16(); k = torch.randn(1, S, Hkv, D, device="cuda").bfloat16(); v = torch.randn(1, S, Hkv, D, device="cuda").bfloat16()
o, lse = ops.flash_attn(q, k, v, causal=True, return_lse=True)
d_o = torch.randn_like(o); dq, dk, dv = torch.empty_like(q), torch.empty_like(k), torch.empty_like(v)
os.environ["VITA_ATTN_BWD_ONLY"] = "dkv"
f = lambda: ops.flash_attn_bwd(q, k, v, o, d_o, lse, dq5=dq, dk=dk, dv=dv)
f(); torch.cuda.synchronize()
ts = []
for _ in range(5):
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record(); f(); b.record(); torch.cuda.synchronize(); ts.append(a.elapsed_time(b))
print(os.environ.get("VITA_HIP_LIB", "default"), "dkv 16K ms:", " ".join(f"{t:.3f}" for t in sorted(ts)))
