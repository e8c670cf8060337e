"""LayerScale-residual backward of the ViT layers (vita_bias_scale_res_bwd) at a 253-frame / 64-frame chunk, 1024 columns."""
import json, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from long_vita_amd import lib, ops
lib.load(allow_build=False)
DEV = "cuda:0"
gen = torch.Generator(device=DEV).manual_seed(1)
for rows, cols in ((259325, 1024), (65600, 1024)):
    x = torch.randn(rows, cols, generator=gen, device=DEV).bfloat16()
    g = torch.randn(rows, cols, generator=gen, device=DEV).bfloat16()
    bias = torch.randn(cols, generator=gen, device=DEV).bfloat16()
    scale = torch.randn(cols, generator=gen, device=DEV).bfloat16()
    db, ds = torch.zeros(cols, device=DEV), torch.zeros(cols, device=DEV)
    f = lambda: ops.bias_scale_residual_bwd(g, x, bias, scale, db, ds)
    for _ in range(3):
        f()
    torch.cuda.synchronize()
    ts = []
    for _ in range(7):
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record(); f(); b.record(); torch.cuda.synchronize()
        ts.append(a.elapsed_time(b))
    t = sorted(ts)[3]
    print(json.dumps(dict(kind="bias_scale_res_bwd", rows=rows, cols=cols, ms=t, tb_per_s=rows * cols * 2 * 3 / t / 1e9)), flush=True)
