"""Developer A / B of the NT GEMM main loop on the decoder shapes (VITA_HIP_LIB selects the build)."""
import os, sys
os.environ.setdefault("VITA_DEBUG", "1")
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT)
import torch
from long_vita_amd import ops
def t(f, n=5):
    f(); torch.cuda.synchronize(); ts = []
    for _ in range(n):
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record(); f(); b.record(); torch.cuda.synchronize(); ts.append(a.elapsed_time(b))
    return sorted(ts)[n // 2]
res = []
for (M, N, K, epi, tag) in [(131072, 5120, 13824, ops.EPI_NONE, "fc2@128K"), (16384, 7168, 5120, ops.EPI_NONE, "qkv@16K"), (131072, 13824, 5120, ops.EPI_SWIGLU, "fc1+swiglu@128K"),
                             (131072, 5120, 5120, ops.EPI_NONE, "o@128K"), (8192, 8192, 8192, ops.EPI_NONE, "8K^3")]:
    a = (torch.randn(M, K, device="cuda") * 0.5).bfloat16(); w = (torch.randn(N * (2 if epi == ops.EPI_SWIGLU else 1), K, device="cuda") * 0.02).bfloat16()
    out = torch.empty(M, N, dtype=torch.bfloat16, device="cuda")
    ms = t(lambda: ops.gemm(a, w, epi, out=out))
    res.append(f"{tag} {ms:.3f} ms {2 * M * N * (2 if epi == ops.EPI_SWIGLU else 1) * K / ms / 1e9:.0f} TF")
    del a, w, out
print(os.environ.get("VITA_HIP_LIB", "default"), " | ".join(res))
