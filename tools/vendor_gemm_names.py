"""Which kernels does the vendor GEMM library run on the decoder shapes?  Run under rocprofv3 --kernel-trace
(profiles/r01_vendor_gemm_kernels.txt); torch.matmul is a yardstick here, not part of the product path."""
import torch
DEV = "cuda"
for (M, N, K) in [(131072, 5120, 13824), (131072, 5120, 5120), (16384, 7168, 5120), (8192, 8192, 8192), (131072, 27648, 5120)]:
    a = (torch.randn(M, K, device=DEV) * 0.5).bfloat16(); w = (torch.randn(N, K, device=DEV) * 0.02).bfloat16()
    out = torch.empty(M, N, dtype=torch.bfloat16, device=DEV)
    for _ in range(3):
        torch.matmul(a, w.t(), out=out)
    torch.cuda.synchronize()
    del a, w, out
print("done")
