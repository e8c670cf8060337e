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
if os.environ.get("ABL_PERIODIC") == "1":        # the DATA of the KVP_ABL = 8 ablation at the real addresses: every head = head 0, rows repeat with period 4096
    q = q[:, :4096, :1].repeat(1, S // 4096, Hq, 1).contiguous(); d_o = d_o[:, :4096, :1].repeat(1, S // 4096, Hq, 1).contiguous()
    lse = lse[:, :1, :4096].repeat(1, Hq, S // 4096).contiguous()
if os.environ.get("ABL_HEADMAJOR") == "1":       # the same values, Q and dO stored head-major ([heads, rows, d]: a tile = 16 KB contiguous) behind strided views
    q = q.transpose(1, 2).contiguous().transpose(1, 2); d_o = d_o.transpose(1, 2).contiguous().transpose(1, 2)
    dq = torch.empty_like(q)
PASS = os.environ.get("ABL_PASS", "dkv")            # "dkv" (attn_bwd_kvp) or "dq" (attn_bwd_dq64)
os.environ["VITA_ATTN_BWD_ONLY"] = PASS
f = lambda: ops.flash_attn_bwd(q, k, v, o, d_o, lse, dq5=dq, dk=dk, dv=dv)
f(); torch.cuda.synchronize()
ts = []
for _ in range(5):
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record(); f(); b.record(); torch.cuda.synchronize(); ts.append(a.elapsed_time(b))
print(os.environ.get("VITA_HIP_LIB", "default"), PASS, "16K ms:", " ".join(f"{t:.3f}" for t in sorted(ts)))
