"""r05: SigLIP-400M's attention (16 heads x 72, 1024 tokens per frame) at head size 96 against the r04 path (zero-padded to 128): the forward
and backward kernels alone, and the whole inference tower (qkv / proj GEMMs carry the padding too).  Writes gpurun_out/r05_siglip_d96.jsonl."""
import json, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT)
import torch
from long_vita_amd import ops, vision
from long_vita_amd.autograd_fns import FlashAttnNonCausalFn

DEV = "cuda"
OUT = os.path.join(ROOT, "gpurun_out", "r05_siglip_d96.jsonl")
os.makedirs(os.path.dirname(OUT), exist_ok=True)


def timed(f, n=10):
    f(); torch.cuda.synchronize()
    ts = []
    for _ in range(n):
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record(); f(); b.record(); torch.cuda.synchronize(); ts.append(a.elapsed_time(b))
    ts.sort()
    return ts[0], ts[len(ts) // 2]


rows = []
B, S, H, hn = 64, 1024, 16, 72                         # 448 / 14 = 32 x 32 patches, no class token
g = torch.Generator(device=DEV).manual_seed(5)
base = [torch.randn(B, S, H, hn, generator=g, device=DEV).bfloat16() for _ in range(4)]
for D in (96, 128):
    q, k, v, d_o = (torch.nn.functional.pad(t, (0, D - hn)) for t in base)
    scale = hn ** -0.5
    best, med = timed(lambda: ops.flash_attn(q, k, v, causal=False, softmax_scale=scale))
    flops = 4.0 * B * H * S * S * hn                       # algorithmic: the TRUE head size
    rows.append({"what": "attention forward", "frames": B, "tokens": S, "heads": H, "head_size": hn, "padded_to": D, "ms_best": best,
                 "ms_median": med, "algorithmic_pflops": flops / best / 1e12})
    qd, kd, vd = (t.clone().requires_grad_(True) for t in (q, k, v))
    out = FlashAttnNonCausalFn.apply(qd, kd, vd, scale)
    best, med = timed(lambda: torch.autograd.grad(out, (qd, kd, vd), d_o, retain_graph=True))
    rows.append({"what": "attention backward (copies + delta + dK/dV + dQ)", "frames": B, "tokens": S, "heads": H, "head_size": hn,
                 "padded_to": D, "ms_best": best, "ms_median": med, "algorithmic_pflops": 2.5 * flops / best / 1e12})
    del qd, kd, vd, out
    # the two backward kernels alone, on the [1, S, frames x heads, D] layout the autograd path builds (query chunk id 1 > key chunk id 0: no mask)
    qh, kh, vh, dh = (t.transpose(0, 1).reshape(1, S, B * H, D).contiguous() for t in (q, k, v, d_o))
    o, lse = ops.flash_attn(qh, kh, vh, causal=False, softmax_scale=scale, return_lse=True)
    dq, dk, dv = torch.empty_like(qh), torch.empty_like(kh), torch.empty_like(vh)
    for part, name in ((ops.ATTN_BWD_DKV, "dK + dV kernel"), (ops.ATTN_BWD_DQ, "dQ kernel")):
        _, _, _, delta = ops.flash_attn_bwd(qh, kh, vh, o, dh, lse, chunk_len=S, q_chunk_gid=[1], kv_chunk_gid=[0], kv_chunk_row=[0], softmax_scale=scale,
                                            dq5=dq, dk=dk, dv=dv, parts=part)
        best, med = timed(lambda: ops.flash_attn_bwd(qh, kh, vh, o, dh, lse, chunk_len=S, q_chunk_gid=[1], kv_chunk_gid=[0], kv_chunk_row=[0],
                                                     softmax_scale=scale, dq5=dq, dk=dk, dv=dv, parts=part, delta=delta))
        units = 1.5 if part == ops.ATTN_BWD_DQ else 2.0            # recomputed S and dP + one (dQ) or two (dK, dV) products, in units of the forward (4 S^2 d)
        rows.append({"what": name, "frames": B, "tokens": S, "heads": H, "head_size": hn, "padded_to": D, "ms_best": best, "ms_median": med,
                     "algorithmic_pflops": units * flops / best / 1e12})
    del qh, kh, vh, dh, o, lse, dq, dk, dv

cfg = vision.VisionConfig.siglip_400m()
imgs = torch.randn(16, 3, cfg.image, cfg.image, generator=g, device=DEV).bfloat16()
orig = vision.VisionConfig.head_dim_pad
for D in (96, 128):
    vision.VisionConfig.head_dim_pad = property(lambda self, D=D: D)
    tower = vision.MegatronVisionModel.random_init(cfg, seed=7, device=DEV)
    with torch.no_grad():
        best, med = timed(lambda: tower.vit(imgs), n=5)
    rows.append({"what": "SigLIP-400M tower (27 layers), 16 frames", "padded_to": D, "ms_best": best, "ms_median": med})
    del tower
vision.VisionConfig.head_dim_pad = orig
with open(OUT, "w") as f:
    for r in rows:
        f.write(json.dumps(r) + "\n"); print(json.dumps(r))
