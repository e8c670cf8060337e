"""LayerNorm backward (ViT block norms, stage 2 trains the encoder) at a 253-frame chunk: 259325 rows x 1024 columns."""
import json, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from long_vita_amd import lib, ops
lib.load(allow_build=False)
DEV = "cuda:0"
g = torch.Generator(device=DEV).manual_seed(1)
for rows, cols in ((259325, 1024), (65600, 1024), (64768, 4096)):
    x = torch.randn(rows, cols, generator=g, device=DEV).bfloat16()
    dy = torch.randn(rows, cols, generator=g, device=DEV).bfloat16()
    w = torch.ones(cols, device=DEV).bfloat16()
    dg, db = torch.zeros(cols, device=DEV), torch.zeros(cols, device=DEV)
    dx = torch.empty_like(x)
    for _ in range(3):
        ops.layernorm_bwd(dy, x, w, 1e-6, dg, db, out=dx)
    torch.cuda.synchronize()
    ts = []
    for _ in range(7):
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record(); ops.layernorm_bwd(dy, x, w, 1e-6, dg, db, out=dx); b.record(); torch.cuda.synchronize()
        ts.append(a.elapsed_time(b))
    t = sorted(ts)[3]
    print(json.dumps(dict(kind="layernorm_bwd", rows=rows, cols=cols, ms=t, tb_per_s=rows * cols * 2 * 3 / t / 1e9)), flush=True)
