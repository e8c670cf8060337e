"""RMSNorm backward A/B (r03): VITA_RMSNORM_BWD = o (r02 kernel) | unset (prefetch + LDS dw image) | w<N> (workgroup per row, N workgroups).
One process per setting (the switch is read once).  Also checks the variant against the r02 kernel's dx / dw on the same inputs
when a reference file from an earlier run exists in /tmp."""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
os.environ.setdefault("VITA_DEBUG", "1")
import torch  # noqa: E402

from long_vita_amd import lib, ops  # noqa: E402

DEV = "cuda:0"
lib.load(allow_build=False)
tag = os.environ.get("VITA_RMSNORM_BWD", "new")


def timeit(fn, warmup=3, iters=9):
    for _ in range(warmup):
        fn()
    torch.cuda.synchronize()
    ts = []
    for _ in range(iters):
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record(); fn(); b.record()
        torch.cuda.synchronize()
        ts.append(a.elapsed_time(b))
    return sorted(ts)[len(ts) // 2]


g = torch.Generator(device=DEV).manual_seed(3)
cols = 5120
w = (1 + 0.1 * torch.randn(cols, generator=g, device=DEV)).bfloat16()
for rows in (16384, 131072):
    x = torch.randn(rows, cols, generator=g, device=DEV).bfloat16()
    dy = (torch.randn(rows, cols, generator=g, device=DEV) * 0.1).bfloat16()
    res = (torch.randn(rows, cols, generator=g, device=DEV) * 0.1).bfloat16()
    dx = torch.empty_like(x)
    dw = torch.zeros(cols, dtype=torch.float32, device=DEV)
    for name, kw in (("dx only", {}), ("dx + dw", dict(dw_acc=dw)), ("dx + dw + residual", dict(dw_acc=dw, residual=res))):
        t = timeit(lambda: ops.rmsnorm_bwd(dy, x, w, 1e-6, out=dx, **kw))
        nbytes = rows * cols * 2 * (3 + ("residual" in kw))
        print(json.dumps(dict(kind="rmsnorm_bwd", variant=tag, rows=rows, what=name, ms=t, tb_per_s=nbytes / t / 1e9)), flush=True)
    if rows == 16384:
        dw.zero_()
        ops.rmsnorm_bwd(dy, x, w, 1e-6, dw_acc=dw, out=dx, residual=res)
        torch.cuda.synchronize()
        ref_file = "/tmp/rmsnorm_bwd_ref.pt"
        if tag == "o":
            torch.save({"dx": dx.cpu(), "dw": dw.cpu()}, ref_file)
        elif os.path.exists(ref_file):
            r = torch.load(ref_file)
            print(json.dumps(dict(kind="rmsnorm_bwd_check", variant=tag, dx_equal=bool(torch.equal(dx.cpu(), r["dx"])),
                                  dx_max_abs=float((dx.cpu().float() - r["dx"].float()).abs().max()),
                                  dw_rel=float((dw.cpu() - r["dw"]).norm() / r["dw"].norm()))), flush=True)
