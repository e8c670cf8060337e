"""Developer aid: error structure of vita_flash_attn_fwd (new vs old kernel) against fp32 torch on the GPU."""
import math, os, sys
os.environ.setdefault("VITA_DEBUG", "1")      # developer switches (VITA_GEMM_*, VITA_ATTN_*) are honoured only with this set
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from long_vita_amd import ops
DEV = "cuda"
S, Hq, Hkv, D = int(os.environ.get("S", 512)), 5, 1, 128
g = torch.Generator(device=DEV).manual_seed(1)
q = torch.randn(1, S, Hq, D, generator=g, device=DEV).bfloat16()
k = torch.randn(1, S, Hkv, D, generator=g, device=DEV).bfloat16()
v = torch.randn(1, S, Hkv, D, generator=g, device=DEV).bfloat16()
sc = torch.einsum("bqhd,bkhd->bhqk", q.float(), k.float().repeat_interleave(Hq // Hkv, 2)) / math.sqrt(D)
sc = sc.masked_fill(torch.triu(torch.ones(S, S, dtype=torch.bool, device=DEV), 1), float("-inf"))
p = torch.softmax(sc, -1)
ref = torch.einsum("bhqk,bkhd->bqhd", p, v.float().repeat_interleave(Hq // Hkv, 2))
# reference with P rounded to bf16 (what every flash kernel multiplies)
m = sc.max(-1, keepdim=True).values
e = torch.exp(sc - m)
ref_b = torch.einsum("bhqk,bkhd->bqhd", e.bfloat16().float(), v.float().repeat_interleave(Hq // Hkv, 2)) / e.sum(-1).permute(0, 2, 1)[..., None]
def rel(a, b): return float((a.float() - b).norm() / b.norm())
for name, env in (("attn64", "1"), ("old", "0")):
    os.environ["VITA_ATTN64"] = env
    o = ops.flash_attn(q, k, v, causal=True).float()
    torch.cuda.synchronize()
    print(name, "rel vs fp32", rel(o, ref), "vs bf16-P", rel(o, ref_b), "bf16-P vs fp32", rel(ref_b, ref))
    err = (o - ref)
    for lo in range(0, S, 64):
        print("  rows %4d-%4d rel %.4f" % (lo, lo + 63, rel(o[:, lo:lo + 64], ref[:, lo:lo + 64])), end="")
        print("   first32 %.4f last32 %.4f" % (rel(o[:, lo:lo + 32], ref[:, lo:lo + 32]), rel(o[:, lo + 32:lo + 64], ref[:, lo + 32:lo + 64])))
    print("  by d-block:", ["%.4f" % rel(o[..., d0:d0 + 32], ref[..., d0:d0 + 32]) for d0 in range(0, D, 32)])
    print("  by head:", ["%.4f" % rel(o[:, :, h], ref[:, :, h]) for h in range(Hq)])
    rn = o.norm(dim=-1) / ref.norm(dim=-1)            # per-row scale: a wrong normaliser shows as a ratio != 1
    print("  row-norm ratio: mean %.5f min %.5f max %.5f" % (float(rn.mean()), float(rn.min()), float(rn.max())))
    cosv = (o * ref).sum(-1) / (o.norm(dim=-1) * ref.norm(dim=-1))
    print("  row cosine: mean %.6f min %.6f" % (float(cosv.mean()), float(cosv.min())))
