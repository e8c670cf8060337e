"""Developer aid: where do two GEMM kernels (VITA_GEMM_KERNEL values) disagree?  Prints the (row, column) pattern of the outliers."""
import os, sys, math
os.environ.setdefault("VITA_DEBUG", "1")      # developer switches (VITA_GEMM_*, VITA_ATTN_*) are honoured only with this set
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from long_vita_amd import ops

M, N, K = (int(x) for x in (sys.argv[1:4] if len(sys.argv) > 3 else (2048, 7168, 5120)))
epi = int(sys.argv[4]) if len(sys.argv) > 4 else 5
g = torch.Generator().manual_seed(1)
a = (torch.randn(M, K, generator=g) * 0.5).bfloat16().cuda()
w = (torch.randn(2 * N if epi == 5 else N, K, generator=g) / math.sqrt(K)).bfloat16().cuda()
bias = (torch.randn(N, generator=g) * 0.1).bfloat16().cuda() if epi in (1, 2, 3, 4) else None
scale = (0.1 + 0.01 * torch.randn(N, generator=g)).bfloat16().cuda() if epi == 4 else None
res = torch.randn(M, N, generator=g).bfloat16().cuda() if epi in (3, 4) else None
outs = {}
for kn in ("w8", "w4"):
    os.environ["VITA_GEMM_KERNEL"] = kn
    outs[kn] = ops.gemm(a, w, epi, bias, scale, res).float()
    torch.cuda.synchronize()
d = (outs["w4"] - outs["w8"]).abs()
tol = 0.02 * outs["w8"].abs().clamp_min(0.05)
bad = (d > tol).nonzero()
print("M N K epi", M, N, K, epi, "rel_l2", float(d.norm() / outs["w8"].norm()), "outliers", bad.shape[0], "of", M * N)
if bad.shape[0]:
    r, c = bad[:, 0], bad[:, 1]
    print("rows % 16 histogram", torch.bincount(r % 16, minlength=16).tolist())
    print("row block (r % 128) // 16", torch.bincount((r % 128) // 16, minlength=8).tolist())
    print("row half (r % 256) // 128", torch.bincount((r % 256) // 128, minlength=2).tolist())
    bn = 128 if epi == 5 else 256
    print("cols % 16 histogram", torch.bincount(c % 16, minlength=16).tolist())
    print("col block (c % wave) // 16", torch.bincount((c % (bn // 2)) // 16, minlength=8).tolist())
    print("tile m", torch.bincount(r // 256).tolist())
    print("tile n", torch.bincount(c // bn).tolist())
    print("first 10", bad[:10].tolist(), [(float(outs['w4'][i, j]), float(outs['w8'][i, j])) for i, j in bad[:5].tolist()])
