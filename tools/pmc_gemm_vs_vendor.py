"""rocprofv3 --pmc driver: the decoder's fc2 GEMM (M = 131072, N = 5120, K = 13824) and qkv at 16K (16384 x 7168 x 5120) through the library's
kernel and through the vendor library (torch.matmul: a yardstick, not part of the product path), 3 launches each."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT)
import torch
from long_vita_amd import ops
SHAPES = [(131072, 5120, 13824), (16384, 7168, 5120)]
if os.environ.get("PMC_SHAPE"):                      # one shape per process: the vendor library runs the SAME kernel and grid on both
    SHAPES = [SHAPES[int(os.environ["PMC_SHAPE"])]]
for (M, N, K) in SHAPES:
    a = (torch.randn(M, K, device="cuda") * 0.5).bfloat16(); w = (torch.randn(N, K, device="cuda") * 0.02).bfloat16()
    out = torch.empty(M, N, dtype=torch.bfloat16, device="cuda")
    for _ in range(3):
        ops.gemm(a, w, ops.EPI_NONE, out=out)
    for _ in range(3):
        torch.matmul(a, w.t(), out=out)
    torch.cuda.synchronize()
    del a, w, out
print("done")
