"""Developer A / B of the TN (weight gradient) and NN (input gradient) GEMM modes (VITA_HIP_LIB selects the build): 16K and config-5 per-rank shapes."""
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
for (T, N, K, tag) in [(16384, 27648, 5120, "fc1@16K"), (16384, 5120, 13824, "fc2@16K"), (16384, 7168, 5120, "qkv@16K"), (32768, 13824, 5120, "cfg5 fc1")]:
    dy = torch.randn(T, N, device="cuda").bfloat16(); x = (torch.randn(T, K, device="cuda") * 0.5).bfloat16()
    w = (torch.randn(N, K, device="cuda") * 0.02).bfloat16()
    out = torch.empty(N, K, dtype=torch.bfloat16, device="cuda"); dx = torch.empty(T, K, dtype=torch.bfloat16, device="cuda")
    ms = t(lambda: ops.gemm_tn(dy, x, out=out)); ms2 = t(lambda: ops.gemm_nn(dy, w, out=dx))
    res.append(f"{tag} wgrad {ms:.3f} ms {2 * T * N * K / ms / 1e9:.0f} TF dgrad {ms2:.3f} ms {2 * T * N * K / ms2 / 1e9:.0f} TF")
    del dy, x, w, out, dx
print(os.environ.get("VITA_HIP_LIB", "default"), " | ".join(res))
