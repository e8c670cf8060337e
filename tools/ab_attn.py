"""Developer A / B of the forward attention kernel (VITA_HIP_LIB selects the build): 40 : 8 heads, d = 128, causal."""
import os, sys
os.environ.setdefault("VITA_DEBUG", "1")
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT)
import torch
from long_vita_amd import ops
res = []
for S, n in ((16384, 7), (32768, 5), (131072, 3)):
    q = torch.randn(1, S, 40, 128, device="cuda").bfloat16(); k = torch.randn(1, S, 8, 128, device="cuda").bfloat16(); v = torch.randn(1, S, 8, 128, device="cuda").bfloat16()
    o = torch.empty_like(q)
    ops.flash_attn(q, k, v, causal=True, out=o); torch.cuda.synchronize(); ts = []
    for _ in range(n):
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record(); ops.flash_attn(q, k, v, causal=True, out=o); b.record(); torch.cuda.synchronize(); ts.append(a.elapsed_time(b))
    ms = sorted(ts)[n // 2]
    res.append(f"S={S}: {ms:.3f} ms {4 * 128 * 40 * S * (S + 1) / 2 / ms / 1e9:.0f} TF")
    del q, k, v, o
print(os.environ.get("VITA_HIP_LIB", "default"), " | ".join(res))
