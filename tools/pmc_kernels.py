"""Tiny driver for rocprofv3 --pmc passes: a few launches of the two dominant kernels.
PMC_S = attention sequence length (default 32768); PMC_GEMM=0 skips the GEMM."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from long_vita_amd import ops
DEV = "cuda"
S = int(os.environ.get("PMC_S", "32768"))
if os.environ.get("PMC_VIT", "0") != "0":            # the ViT's attention: 253 frames x 1025 tokens, 16 heads x 64, non-causal
    B = int(os.environ.get("PMC_VIT_FRAMES", "253"))
    q, k, v = (torch.randn(B, 1025, 16, 64, device=DEV).bfloat16() for _ in range(3))
    o = torch.empty_like(q)
    for _ in range(3):
        ops.flash_attn(q, k, v, causal=False, out=o)
    torch.cuda.synchronize()
    print("done")
    sys.exit(0)
q = torch.randn(1, S, 40, 128, device=DEV).bfloat16(); k = torch.randn(1, S, 8, 128, device=DEV).bfloat16()
v = torch.randn(1, S, 8, 128, device=DEV).bfloat16(); o = torch.empty_like(q)
do_gemm = os.environ.get("PMC_GEMM", "1") != "0"
do_bwd = os.environ.get("PMC_BWD", "0") != "0"          # the attention backward (delta + dQ + dK/dV kernels) at the same S instead of the forward
if do_gemm:
    M = 131072 if S >= 131072 else 16384      # the decoder's fc1 + SwiGLU GEMM at the same token count
    a = (torch.randn(M, 5120, device=DEV) * 0.5).bfloat16(); w = (torch.randn(2 * 13824, 5120, device=DEV) * 0.02).bfloat16()
    c = torch.empty(M, 13824, dtype=torch.bfloat16, device=DEV)
if do_bwd:
    o, lse = ops.flash_attn(q, k, v, causal=True, return_lse=True)
    d_o = torch.randn_like(o); dq, dk, dv = torch.empty_like(q), torch.empty_like(k), torch.empty_like(v)
for _ in range(3):
    if do_bwd:
        ops.flash_attn_bwd(q, k, v, o, d_o, lse, dq5=dq, dk=dk, dv=dv)
        continue
    ops.flash_attn(q, k, v, causal=True, out=o)
    if do_gemm:
        ops.gemm(a, w, ops.EPI_SWIGLU, out=c)
torch.cuda.synchronize()
print("done")
