"""Secondary measurement: flash attention forward + backward on one 16K row, unpacked vs packed into
8 samples of 2K (block-diagonal causal): the packed launch should cost about the sum of the samples' triangles."""
import json, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from long_vita_amd import lib, ops
lib.load(allow_build=False)
DEV = "cuda"
S, Hq, Hkv, D = 16384, 40, 8, 128
q = torch.randn(1, S, Hq, D, device=DEV).bfloat16(); k = torch.randn(1, S, Hkv, D, device=DEV).bfloat16()
v = torch.randn(1, S, Hkv, D, device=DEV).bfloat16(); d_o = torch.randn(1, S, Hq, D, device=DEV).bfloat16()
cu = torch.arange(0, S + 1, 2048, dtype=torch.int32, device=DEV)
seg_start, seg_end = ops.segments_from_cu_seqlens(cu, S)


def timed(fn, n=5):
    fn(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n


res = {}
for tag, kw, kwb in (("unpacked", {}, {}), ("packed_8x2k", {"seg_start": seg_start}, {"seg_start": seg_start, "seg_end": seg_end})):
    o, lse = ops.flash_attn(q, k, v, causal=True, return_lse=True, **kw)
    res[tag + "_fwd_ms"] = timed(lambda: ops.flash_attn(q, k, v, causal=True, return_lse=True, **kw))
    res[tag + "_bwd_ms"] = timed(lambda: ops.flash_attn_bwd(q, k, v, o, d_o, lse, **kwb))
print(json.dumps({"what": "attention fwd / bwd, 16K row, 40:8 heads, d=128", **res}))
