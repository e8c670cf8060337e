import os, sys, math, torch
sys.path.insert(0, "/root/repo")
from long_vita_amd import ops
from oracle import attention as oattn
def ref(q, k, v):
    o = oattn.core_attention(q.transpose(0, 1).float(), k.transpose(0, 1).float(), v.transpose(0, 1).float(), False)
    B, S, H, D = q.shape
    return o.view(S, B, H, D).transpose(0, 1)
for (B, S, H) in [(1, 64, 1), (1, 128, 1), (1, 192, 1), (1, 256, 1), (1, 320, 1), (1, 65, 1), (2, 1025, 2)]:
    g = torch.Generator().manual_seed(S)
    q = torch.randn(B, S, H, 64, generator=g).bfloat16(); k = torch.randn(B, S, H, 64, generator=g).bfloat16(); v = torch.randn(B, S, H, 64, generator=g).bfloat16()
    r = ref(q, k, v)
    out = ops.flash_attn(q.cuda(), k.cuda(), v.cuda(), causal=False).cpu().float()
    e = (out - r)
    print(f"B{B} S{S} H{H}: rel {float(e.norm() / r.norm()):.4f}", " by row block of 32:", [round(float(e[0, i:i + 32].norm() / r[0, i:i + 32].norm()), 3) for i in range(0, min(S, 256), 32)],
          " by d half:", [round(float(e[..., j:j + 32].norm() / r[..., j:j + 32].norm()), 3) for j in (0, 32)])
    # V = identity-like probe: out = P @ V with V one-hot in d -> reveals key mixing
    if S == 64:
        v2 = torch.zeros(1, 64, 1, 64); v2[0, torch.arange(64), 0, torch.arange(64)] = 1.0
        q2 = torch.zeros(1, 64, 1, 64).bfloat16(); k2 = torch.zeros(1, 64, 1, 64).bfloat16()
        o2 = ops.flash_attn(q2.cuda(), k2.cuda(), v2.bfloat16().cuda(), causal=False).cpu().float()
        print("  uniform P, V = I: row 0 (want all 1/64 = 0.0156):", o2[0, 0, 0, :8].tolist(), "min", float(o2.min()), "max", float(o2.max()))
        # K probe: scores depend on key index: k[j] = e_(j % 64) * s, q = ones
        k3 = torch.zeros(1, 64, 1, 64); k3[0, torch.arange(64), 0, torch.arange(64)] = 8.0 * (torch.arange(64) == 5).float()
        q3 = torch.ones(1, 64, 1, 64)
        o3 = ops.flash_attn(q3.bfloat16().cuda(), k3.bfloat16().cuda(), v2.bfloat16().cuda(), causal=False).cpu().float()
        print("  key 5 boosted: argmax d of row 0:", int(o3[0, 0, 0].argmax()), "value", float(o3[0, 0, 0].max()))
