"""Packed-sample attention (stage 2's `--reset-position-ids` packing), forward + backward at a 32K pack,
40 : 8 heads, d = 128: the 64-row kernels' packed variants (default) against the r01 kernels (VITA_ATTN64=0 VITA_ATTN_BWD64=0).
    python tools/bench_packed_attn.py            (run once per setting; appends to gpurun_out/r03_packed_attn.jsonl)"""
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
OUT = os.path.join(ROOT, "gpurun_out")
os.makedirs(OUT, exist_ok=True)


def timeit(fn, warmup=2, iters=5):
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


S, Hq, Hkv, D = 32768, 40, 8, 128
g = torch.Generator(device=DEV).manual_seed(1)
q = torch.randn(1, S, Hq, D, generator=g, device=DEV).bfloat16()
k = torch.randn(1, S, Hkv, D, generator=g, device=DEV).bfloat16()
v = torch.randn(1, S, Hkv, D, generator=g, device=DEV).bfloat16()
d_o = torch.randn(1, S, Hq, D, generator=g, device=DEV).bfloat16()
for name, cu in (("8 samples of 1.3K .. 9.7K", [0, 1301, 5000, 9111, 18811, 20000, 24001, 29999]), ("1 sample", [0]),
                 ("64 samples of 512", list(range(0, S, 512)))):
    cu_t = torch.tensor(cu, dtype=torch.int32, device=DEV)
    seg_start, seg_end = ops.segments_from_cu_seqlens(cu_t, S)
    ends = cu[1:] + [S]
    pairs = sum((b - a) * (b - a + 1) / 2 for a, b in zip(cu, ends))
    unit = 2.0 * D * Hq * pairs
    out, lse = ops.flash_attn(q, k, v, causal=True, return_lse=True, seg_start=seg_start)
    t_f = timeit(lambda: ops.flash_attn(q, k, v, causal=True, out=out, seg_start=seg_start))
    t_b = timeit(lambda: ops.flash_attn_bwd(q, k, v, out, d_o, lse, seg_start=seg_start, seg_end=seg_end))
    rec = dict(kind="packed_attn", pack=name, S=S, heads=f"{Hq}:{Hkv}", attn64=os.environ.get("VITA_ATTN64", "1"),
               bwd64=os.environ.get("VITA_ATTN_BWD64", "1"), fwd_ms=t_f, bwd_ms=t_b, fwd_algorithmic_tflops=2 * unit / t_f / 1e9,
               bwd_algorithmic_tflops=5 * unit / t_b / 1e9)
    print(json.dumps(rec), flush=True)
    open(os.path.join(OUT, "r03_packed_attn.jsonl"), "a").write(json.dumps(rec) + "\n")
