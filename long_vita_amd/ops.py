"""Tensor-level entry points over the C ABI (include/vita_hip.h).

PyTorch supplies device memory and the current HIP stream only; all arithmetic happens in
libvita_hip.so.  Every function raises if its inputs are not on a HIP device — there is no CPU
fallback and nothing here imports ``oracle/``.
"""
from __future__ import annotations

import ctypes as C
import math
import os
from typing import Optional, Sequence

import torch

from . import lib as _L
from .lib import (EPI_BIAS, EPI_BIAS2_GELU, EPI_BIAS2_GELU_TANH, EPI_BIAS2_RES, EPI_BIAS_GELU, EPI_BIAS_SCALE_RES, EPI_NONE,  # noqa: F401
                  EPI_RESIDUAL, EPI_SWIGLU, AttnParams)

BF16 = torch.bfloat16


def _stream() -> int:
    return torch.cuda.current_stream().cuda_stream


def _dev(t: torch.Tensor, name: str, dtype=None) -> int:
    if not isinstance(t, torch.Tensor) or not t.is_cuda:
        raise RuntimeError(f"{name} must be a HIP device tensor (the product path has no CPU fallback)")
    if dtype is not None and t.dtype != dtype:
        raise ValueError(f"{name} must be {dtype}, got {t.dtype}")
    return t.data_ptr()


def _opt(t: Optional[torch.Tensor], name: str, dtype=None) -> Optional[int]:
    return None if t is None else _dev(t, name, dtype)


# ------------------------------------------------------------------------------------------------
# norms
# ------------------------------------------------------------------------------------------------
def rmsnorm(x: torch.Tensor, weight: torch.Tensor, eps: float = 1e-6, out: Optional[torch.Tensor] = None,
            rstd: Optional[torch.Tensor] = None) -> torch.Tensor:
    """RMSNorm.forward — M/core/transformer/custom_layers/transformer_engine.py:74-79."""
    cols = x.shape[-1]
    xc = x if x.is_contiguous() else x.contiguous()
    rows = xc.numel() // cols
    y = torch.empty_like(xc) if out is None else out
    if rows == 0:
        return y
    _L.check(_L.load().vita_rmsnorm_fwd(_dev(xc, "x", BF16), _dev(weight, "weight", BF16), _dev(y, "out", BF16),
                                        _opt(rstd, "rstd", torch.float32), rows, cols, float(eps), _stream()),
             "vita_rmsnorm_fwd")
    return y


def layernorm(x: torch.Tensor, weight: torch.Tensor, bias: Optional[torch.Tensor], eps: float,
              out: Optional[torch.Tensor] = None) -> torch.Tensor:
    cols = x.shape[-1]
    xc = x if x.is_contiguous() else x.contiguous()
    rows = xc.numel() // cols
    y = torch.empty_like(xc) if out is None else out
    if rows == 0:
        return y
    _L.check(_L.load().vita_layernorm_fwd(_dev(xc, "x", BF16), _dev(weight, "weight", BF16), _opt(bias, "bias", BF16),
                                          _dev(y, "out", BF16), rows, cols, float(eps), _stream()),
             "vita_layernorm_fwd")
    return y


# ------------------------------------------------------------------------------------------------
# RoPE
# ------------------------------------------------------------------------------------------------
def rope_inv_freq(dim: int, base: float, device) -> torch.Tensor:
    """inv_freq exactly as RotaryEmbedding.__init__ (rotary_pos_embedding.py:74-80) builds it
    (64 floats; computed with torch on the host so the bit pattern equals the reference's)."""
    inv = 1.0 / (base ** (torch.arange(0, dim, 2, dtype=torch.float32) / dim))
    return inv.to(device)


def rope_table(positions: torch.Tensor, inv_freq: torch.Tensor):
    """cos/sin (bf16 [n, dim/2]) for int64 global positions."""
    pos = positions.reshape(-1)
    if pos.dtype != torch.int64:
        raise ValueError("positions must be int64")
    n, half = pos.numel(), inv_freq.numel()
    cos = torch.empty((n, half), dtype=BF16, device=pos.device)
    sin = torch.empty_like(cos)
    _L.check(_L.load().vita_rope_table(_dev(pos.contiguous(), "positions"), _dev(inv_freq, "inv_freq", torch.float32),
                                       _dev(cos, "cos"), _dev(sin, "sin"), n, half, _stream()), "vita_rope_table")
    return cos, sin


def rope_cos_sin(freqs: torch.Tensor):
    """Megatron's fp32 angles `freqs` [s, 1, 1, dim] (RotaryEmbedding.forward: cat(freqs, freqs)) -> cos / sin bf16 [s, dim/2]."""
    if freqs.dtype != torch.float32 or freqs.dim() != 4 or freqs.shape[1] != 1 or freqs.shape[2] != 1:
        raise ValueError("freqs must be fp32 [s, 1, 1, dim]")
    f = freqs if freqs.is_contiguous() else freqs.contiguous()
    n, dim = f.shape[0], f.shape[3]
    cos = torch.empty((n, dim // 2), dtype=BF16, device=f.device)
    sin = torch.empty_like(cos)
    _L.check(_L.load().vita_rope_cos_sin(_dev(f, "freqs", torch.float32), dim, _dev(cos, "cos"), _dev(sin, "sin"), n, dim // 2,
                                         _stream()), "vita_rope_cos_sin")
    return cos, sin


def rope_apply_(t: torch.Tensor, cos: torch.Tensor, sin: torch.Tensor, sign: int = 1) -> torch.Tensor:
    """In-place apply_rotary_pos_emb_bshd on t [rows, heads, d] (any row/head stride, d contiguous)."""
    if t.dim() != 3 or t.stride(2) != 1:
        raise ValueError("t must be [rows, heads, head_dim] with contiguous head_dim")
    rows, heads, d = t.shape
    _L.check(_L.load().vita_rope_apply(_dev(t, "t", BF16), rows, heads, d, t.stride(0), t.stride(1),
                                       _dev(cos, "cos", BF16), _dev(sin, "sin", BF16), sign, _stream()),
             "vita_rope_apply")
    return t


def cp_kv_split(kv_groups: int, n_heads: int, s_local: int) -> int:
    """K / V all-gather messages (= attention launches) per layer under context parallelism.  Gather j + 1 runs under the attention of
    split j, gather 0 under the attention over the rank's own chunks — so more splits hide more of the exchange, but every split is
    a launch over n_heads / split heads x s_local / 256 row tiles, one 512-register workgroup per CU, and short launches waste their
    last round.  Measured on one rank of the 128K prefill at CP = 8 (S_l = 16384, 2560 workgroups per layer; own chunks first, one
    stream per split; tools/bench_cp8_rank_attn.py, profiles/r03_cp8_rank_attn.jsonl): 1 message 18.0-18.3 ms per layer, 2 messages
    18.8-19.0, 4 messages 18.9-19.9 (without the streams 20.3-26.5) — against 0.3-1 ms of exchange per layer that the own-chunk
    attention (2.2 ms) covers at any split.  Rule: split as finely as every launch keeps >= 10 full rounds (2560 workgroups);
    problems too small for one such launch keep the finest split (nothing fills the chip there, the launches overlap on their
    streams).  VITA_CP_KV_SPLIT=n overrides with AT MOST n messages (a deployment knob: the right value depends on the node's
    all-gather bandwidth)."""
    finest = 4 if kv_groups % 4 == 0 else (2 if kv_groups % 2 == 0 else 1)
    env = os.environ.get("VITA_CP_KV_SPLIT")
    if env:
        n = int(env)
        if n < 1:
            raise ValueError(f"VITA_CP_KV_SPLIT={env}: a positive number of messages")
        while kv_groups % n:                 # at most that many: a tensor-parallel rank holds fewer kv heads
            n -= 1
        return n
    n, wgs = finest, n_heads * max(s_local // 256, 1)
    while n > 1 and wgs >= 2560 and wgs // n < 2560:
        n //= 2
    return n


def rope_qkv_(mixed_qkv: torch.Tensor, groups: int, q_per_group: int, head_dim: int, cos: torch.Tensor,
              sin: torch.Tensor, kv_out: Optional[torch.Tensor] = None, kv_split: int = 1) -> torch.Tensor:
    """Rotate Q/K of Megatron's mixed QKV [rows, groups*(qpg+2)*d] in place; optionally pack
    rotated K and V into kv_out [kv_split, 2, rows, groups / kv_split, d]."""
    if not mixed_qkv.is_contiguous():
        raise ValueError("mixed_qkv must be contiguous")
    rows = mixed_qkv.numel() // (groups * (q_per_group + 2) * head_dim)
    if kv_out is not None and (not kv_out.is_contiguous() or kv_out.numel() != 2 * rows * groups * head_dim):
        raise ValueError("kv_out must be contiguous [2, rows, groups, head_dim]")
    _L.check(_L.load().vita_rope_qkv_fwd(_dev(mixed_qkv, "mixed_qkv", BF16), rows, groups, q_per_group, head_dim,
                                         _dev(cos, "cos", BF16), _dev(sin, "sin", BF16), _opt(kv_out, "kv_out", BF16),
                                         int(kv_split), _stream()), "vita_rope_qkv_fwd")
    return mixed_qkv


# ------------------------------------------------------------------------------------------------
# rows
# ------------------------------------------------------------------------------------------------
def _err_flag(device) -> torch.Tensor:
    return torch.zeros(1, dtype=torch.int32, device=device)


def row_gather(src: torch.Tensor, idx: torch.Tensor, out: Optional[torch.Tensor] = None,
               check_bounds: bool = True) -> torch.Tensor:
    """out[i] = src[idx[i]] over the first dimension (rows are the flattened trailing dims)."""
    if not src.is_contiguous():
        raise ValueError("src must be contiguous")
    if idx.dtype != torch.int64:
        raise ValueError("idx must be int64")
    idx = idx.reshape(-1).contiguous()
    n, cols = idx.numel(), src.numel() // max(src.shape[0], 1)
    y = torch.empty((n,) + tuple(src.shape[1:]), dtype=src.dtype, device=src.device) if out is None else out
    if n == 0:
        return y
    flag = _err_flag(src.device) if check_bounds else None
    _L.check(_L.load().vita_row_gather(_dev(src, "src"), src.shape[0], _dev(idx, "idx"), _dev(y, "out"), n, cols,
                                       src.element_size(), _opt(flag, "flag"), _stream()), "vita_row_gather")
    if check_bounds and int(flag.item()):
        raise IndexError("vita_row_gather: index out of range")
    return y


def row_scatter_(dst: torch.Tensor, dst_idx: torch.Tensor, src: torch.Tensor,
                 src_idx: Optional[torch.Tensor] = None, check_bounds: bool = True) -> torch.Tensor:
    """dst[dst_idx[i]] = src[src_idx[i] if src_idx is not None else i] (rows = flattened trailing dims)."""
    if not dst.is_contiguous() or not src.is_contiguous():
        raise ValueError("dst and src must be contiguous")
    dst_idx = dst_idx.reshape(-1).contiguous()
    if src_idx is not None:
        src_idx = src_idx.reshape(-1).contiguous()
    n = dst_idx.numel()
    if n == 0:                                  # nothing selected (an all-False logit mask on this rank): dst as it is
        return dst
    cols = dst.numel() // max(dst.shape[0], 1)
    if src.numel() // max(src.shape[0], 1) != cols or src.dtype != dst.dtype:
        raise ValueError("row width / dtype mismatch")
    flag = _err_flag(dst.device) if check_bounds else None
    _L.check(_L.load().vita_row_scatter(_dev(src, "src"), src.shape[0], _opt(src_idx, "src_idx", torch.int64),
                                        _dev(dst, "dst"), dst.shape[0], _dev(dst_idx, "dst_idx", torch.int64), n, cols,
                                        dst.element_size(), _opt(flag, "flag"), _stream()), "vita_row_scatter")
    if check_bounds and int(flag.item()):
        raise IndexError("vita_row_scatter: index out of range")
    return dst


def mask_to_index(mask: torch.Tensor) -> torch.Tensor:
    """Positions of the True entries of a flattened bool mask, in order (index form of
    torch.masked_select, M/core/tensor_parallel/layers.py:348,407)."""
    m = mask.reshape(-1)
    if m.dtype == torch.bool:
        m = m.view(torch.uint8)
    if m.dtype != torch.uint8:
        raise ValueError("mask must be bool / uint8")
    m = m.contiguous()
    n = m.numel()
    idx = torch.empty(n, dtype=torch.int64, device=m.device)
    cnt = torch.zeros(1, dtype=torch.int64, device=m.device)
    _L.check(_L.load().vita_mask_to_index(_dev(m, "mask"), n, _dev(idx, "idx"), _dev(cnt, "count"), _stream()),
             "vita_mask_to_index")
    return idx[: int(cnt.item())]


def cp_index_remap(indices_s: torch.Tensor, seq_len: int, cp_size: int, cp_rank: int):
    s = indices_s.contiguous()
    hit = torch.empty(s.shape, dtype=torch.uint8, device=s.device)
    local = torch.empty(s.shape, dtype=torch.int64, device=s.device)
    _L.check(_L.load().vita_cp_index_remap(_dev(s, "indices_s", torch.int64), s.numel(), seq_len, cp_size, cp_rank,
                                           _dev(hit, "hit"), _dev(local, "local"), _stream()), "vita_cp_index_remap")
    return hit, local


def rows_any(mask: torch.Tensor) -> torch.Tensor:
    m = mask.contiguous()
    rows, cols = m.shape
    out = torch.empty(rows, dtype=torch.uint8, device=m.device)
    _L.check(_L.load().vita_rows_any(_dev(m, "mask", torch.uint8), rows, cols, _dev(out, "out"), _stream()),
             "vita_rows_any")
    return out


def index_inverse(idx: torch.Tensor, size: int) -> torch.Tensor:
    inv = torch.full((size,), -1, dtype=torch.int64, device=idx.device)
    _L.check(_L.load().vita_index_inverse(_dev(idx.contiguous(), "idx", torch.int64), idx.numel(), _dev(inv, "inv"),
                                          _stream()), "vita_index_inverse")
    return inv


def cp_src_tgt(hit_idx, tok_per_img, img_rank, indices_b, local_pos):
    n = hit_idx.numel()
    outs = [torch.empty(n, dtype=torch.int64, device=hit_idx.device) for _ in range(4)]
    _L.check(_L.load().vita_cp_src_tgt(_dev(hit_idx, "hit_idx", torch.int64), n, tok_per_img,
                                       _dev(img_rank, "img_rank", torch.int64),
                                       _dev(indices_b.contiguous(), "indices_b", torch.int64),
                                       _dev(local_pos, "local_pos", torch.int64),
                                       *[_dev(o, "out") for o in outs], _stream()), "vita_cp_src_tgt")
    return outs


# ------------------------------------------------------------------------------------------------
# GEMM
# ------------------------------------------------------------------------------------------------
def gemm(a: torch.Tensor, w: torch.Tensor, epilogue: int = EPI_NONE, bias: Optional[torch.Tensor] = None,
         scale: Optional[torch.Tensor] = None, residual: Optional[torch.Tensor] = None,
         out: Optional[torch.Tensor] = None) -> torch.Tensor:
    """out[M, N] = epilogue(a[M, K] @ w[N(or 2N), K]^T) — see VITA_EPI_* in include/vita_hip.h.
    `a` may have a row stride (a.stride(0) >= K); w is [rows, K] contiguous like nn.Linear.weight."""
    if a.dim() != 2 or w.dim() != 2 or a.stride(1) != 1 or w.stride(1) != 1:
        raise ValueError("a and w must be 2-D with contiguous last dim")
    M, K = a.shape
    wn, wk = w.shape
    if wk != K:
        raise RuntimeError(f"supplied weight's shape is {tuple(w.shape)}, K = {K} expected")  # layers.py:849-853
    N = wn // 2 if epilogue == EPI_SWIGLU else wn
    y = torch.empty((M, N), dtype=BF16, device=a.device) if out is None else out
    if y.shape != (M, N) or y.stride(1) != 1:
        raise ValueError("out has wrong shape")
    if M == 0:                                  # an empty selection (a logit mask with no True on this rank): nothing to launch
        return y
    if K % 64:
        # the kernels step the contraction in 64s; a contraction that is not a multiple (SigLIP's FFN 4304 = 67 x 64 + 16,
        # M/pretrain_long_vita.py:268-307) runs on zero-padded copies of both operands: exact, the padding adds zeros to every sum
        kp = (K + 63) // 64 * 64
        a = torch.nn.functional.pad(a, (0, kp - K))
        w = _padded_weight(w, kp)
        K = kp
    r_ptr, ldr = None, 0
    if residual is not None:
        if residual.shape != (M, N) or residual.stride(1) != 1:
            raise ValueError("residual has wrong shape")
        r_ptr, ldr = _dev(residual, "residual", BF16), residual.stride(0)
    _L.check(_L.load().vita_gemm_bf16(_dev(a, "a", BF16), a.stride(0), _dev(w, "w", BF16), w.stride(0),
                                      _dev(y, "out", BF16), y.stride(0), M, N, K, epilogue, _opt(bias, "bias", BF16),
                                      _opt(scale, "scale", BF16), r_ptr, ldr, _stream()), "vita_gemm_bf16")
    return y


_PAD_CACHE: dict = {}


def _padded_weight(w: torch.Tensor, kp: int) -> torch.Tensor:
    """The zero-padded copy of a weight whose contraction length is not a multiple of 64 (SigLIP's fc2: K = 4304).  Cached ONLY for
    FROZEN `torch.nn.Parameter`s (requires_grad False: inference, `--vision-model-freeze`) — keyed by storage identity and shape and
    validated by torch's in-place version counter, which `load_state_dict` / `copy_` bump.  A trainable weight is padded on every call:
    optimizers write through `param.data` (Megatron's Float16Optimizer._copy_main_params_to_model_params, the distributed optimizer's
    param-buffer all-gather, apex multi-tensor kernels) and `.data` writes do NOT move the version counter, so a cached copy would go
    stale after the first step.  Code that rewrites a frozen weight through `.data` calls `invalidate_padded_weights()`.
    Anything else (activations, the transposed weight of a dgrad) is padded per call.  The cache drops its oldest entry beyond 128."""
    if not isinstance(w, torch.nn.Parameter) or w.requires_grad:
        return torch.nn.functional.pad(w.detach(), (0, kp - w.shape[1]))
    key = (w.data_ptr(), tuple(w.shape), w.stride(0), w.device.index)
    hit = _PAD_CACHE.get(key)
    if hit is not None and hit[0] == w._version:
        return hit[1]
    padded = torch.nn.functional.pad(w.detach(), (0, kp - w.shape[1]))
    if len(_PAD_CACHE) >= 128:
        _PAD_CACHE.pop(next(iter(_PAD_CACHE)))
    _PAD_CACHE[key] = (w._version, padded)
    return padded


def invalidate_padded_weights() -> None:
    """Drop every cached padded weight (after frozen weights were rewritten through `.data`, which torch's version counter cannot see)."""
    _PAD_CACHE.clear()


def gemm_tn_ok(a_t: torch.Tensor, w_t: torch.Tensor) -> bool:
    """Shapes vita_gemm_bf16_tn takes: out dims multiples of 256, contraction a multiple of 64, 16-byte aligned rows."""
    return (a_t.dim() == 2 and w_t.dim() == 2 and a_t.shape[0] == w_t.shape[0] and a_t.stride(1) == 1 and w_t.stride(1) == 1
            and a_t.shape[1] % 256 == 0 and w_t.shape[1] % 256 == 0 and a_t.shape[0] % 64 == 0 and a_t.stride(0) % 8 == 0
            and w_t.stride(0) % 8 == 0 and a_t.shape[0] > 0)


def gemm_nn_ok(a: torch.Tensor, w: torch.Tensor) -> bool:
    """Shapes vita_gemm_bf16_nn takes (a [M, K] row-major, w [K, N] contraction-major): whole 256 x 256 output tiles, K a multiple of 64."""
    if os.environ.get("VITA_DEBUG") and os.environ.get("VITA_DGRAD_NN") == "0":        # developer A / B switch: the transposing path
        return False
    return (a.dim() == 2 and w.dim() == 2 and a.shape[1] == w.shape[0] and a.stride(1) == 1 and w.stride(1) == 1
            and a.shape[0] % 256 == 0 and w.shape[1] % 256 == 0 and a.shape[1] % 64 == 0 and a.stride(0) % 8 == 0
            and w.stride(0) % 8 == 0 and a.shape[0] > 0 and a.dtype == BF16 and w.dtype == BF16)


def gemm_nn(a: torch.Tensor, w: torch.Tensor, out: Optional[torch.Tensor] = None) -> torch.Tensor:
    """out[M, N] = a[M, K] @ w[K, N] — the dgrad GEMM `grad_output.matmul(weight)` (M/core/tensor_parallel/layers.py:444,453) with the
    weight [out_features, in_features] as the forward holds it: no vita_transpose_bf16 pass (r04)."""
    M, K = a.shape
    N = w.shape[1]
    y = torch.empty((M, N), dtype=BF16, device=a.device) if out is None else out
    _L.check(_L.load().vita_gemm_bf16_nn(_dev(a, "a", BF16), a.stride(0), _dev(w, "w", BF16), w.stride(0), _dev(y, "out", BF16), y.stride(0),
                                         M, N, K, _stream()), "vita_gemm_bf16_nn")
    return y


def tn_splits(M: int, N: int, K: int) -> int:
    """How many ranges to cut the contraction of a TN (weight-gradient) GEMM into.  The kernel gives one 256 x 256 output tile to one
    workgroup; a gradient whose output is a few tiles (ViT linears: 16 - 64 tiles, the decoder's qkv / proj at config 5's TP-halved widths:
    200 - 280) leaves most of the 256 CUs idle or runs a nearly empty last round.  Cost model in units of one K tile of one round
    (~1.5 us): rounds(tiles x S) x K tiles per split + the fp32 partial write / read of the reduction at ~3 TB/s; split only for >= 10 %."""
    tiles, nk = (M // 256) * (N // 256), K // 64
    unit = 1.5e-6

    def cost(S):
        per = -(-nk // S)
        red = 0.0 if S == 1 else (S * M * N * 8 + M * N * 2) / 3e12 / unit + 4
        return -(-tiles * S // 256) * per + red
    best_s, best = 1, cost(1)
    for S in (2, 3, 4, 6, 8, 12, 16):
        per = -(-nk // S)
        if per < 8 or nk - per * (S - 1) < 1 or S * M * N * 4 > (2 << 30):
            continue
        c = cost(S)
        if c < 0.9 * best and c < 0.9 * cost(1):
            best_s, best = S, c
    return best_s


def gemm_tn(a_t: torch.Tensor, w_t: torch.Tensor, out: Optional[torch.Tensor] = None, splits: Optional[int] = None) -> torch.Tensor:
    """out[M, N] = a_t[K, M]^T @ w_t[K, N] — both operands contraction-major (the wgrad GEMM: a_t = grad_output [tokens, out],
    w_t = total_input [tokens, in]; M/core/tensor_parallel/layers.py:522-523) with no transposed copies.  splits: None = tn_splits'
    choice (r04: split-K for gradients with few output tiles), 1 = one workgroup per tile over the whole contraction."""
    K, M = a_t.shape
    N = w_t.shape[1]
    y = torch.empty((M, N), dtype=BF16, device=a_t.device) if out is None else out
    S = tn_splits(M, N, K) if splits is None else int(splits)
    if S <= 1:
        _L.check(_L.load().vita_gemm_bf16_tn(_dev(a_t, "a_t", BF16), a_t.stride(0), _dev(w_t, "w_t", BF16), w_t.stride(0),
                                             _dev(y, "out", BF16), y.stride(0), M, N, K, _stream()), "vita_gemm_bf16_tn")
        return y
    ws = torch.empty((S, M, N), dtype=torch.float32, device=a_t.device)
    _L.check(_L.load().vita_gemm_bf16_tn_splitk(_dev(a_t, "a_t", BF16), a_t.stride(0), _dev(w_t, "w_t", BF16), w_t.stride(0),
                                                _dev(y, "out", BF16), y.stride(0), M, N, K, S, _dev(ws, "workspace", torch.float32),
                                                _stream()), "vita_gemm_bf16_tn_splitk")
    return y


def colsum(x: torch.Tensor) -> torch.Tensor:
    """fp32 [cols] = sum over the rows of x [rows, cols] bf16 (row stride allowed): grad_bias = grad_output.sum(dim=0) (layers.py:524)."""
    if x.dim() != 2 or x.stride(1) != 1:
        raise ValueError("x must be 2-D with a unit inner stride")
    rows, cols = x.shape
    out = torch.zeros(cols, dtype=torch.float32, device=x.device)
    if rows:
        h = _L.load()
        # the ordered form (ABI 18): row-block partials in a workspace, added in block order — two runs give the same bits (the atomic form
        # vita_colsum_bf16 adds in arrival order)
        ws = torch.empty(max(1, h.vita_colsum_workspace_bytes(rows, cols) // 4), dtype=torch.float32, device=x.device)
        _L.check(h.vita_colsum_bf16_ordered(_dev(x, "x", BF16), x.stride(0), _dev(out, "out", torch.float32), rows, cols,
                                            _dev(ws, "workspace", torch.float32), _stream()), "vita_colsum_bf16_ordered")
    return out


def gemm_skinny(a: torch.Tensor, w: torch.Tensor, out_f32: bool = False) -> torch.Tensor:
    """Logits-masked head GEMM for <= 16 selected rows."""
    M, K = a.shape
    N = w.shape[0]
    if w.shape[1] != K:
        raise RuntimeError(f"supplied weight's shape is {tuple(w.shape)}, K = {K} expected")
    y = torch.empty((M, N), dtype=torch.float32 if out_f32 else BF16, device=a.device)
    if M == 0:
        return y
    _L.check(_L.load().vita_gemm_skinny_bf16(_dev(a, "a", BF16), a.stride(0), _dev(w, "w", BF16), w.stride(0),
                                             _dev(y, "out"), y.stride(0), M, N, K, int(out_f32), _stream()),
             "vita_gemm_skinny_bf16")
    return y


# ------------------------------------------------------------------------------------------------
# attention
# ------------------------------------------------------------------------------------------------
def flash_attn(q: torch.Tensor, k: torch.Tensor, v: torch.Tensor, *, causal: bool, softmax_scale: Optional[float] = None,
               chunk_len: Optional[int] = None, q_chunk_gid: Optional[Sequence[int]] = None,
               kv_chunk_gid: Optional[Sequence[int]] = None, kv_chunk_row: Optional[Sequence[int]] = None,
               out: Optional[torch.Tensor] = None, return_lse: bool = False, seg_start: Optional[torch.Tensor] = None,
               lse_out: Optional[torch.Tensor] = None):
    """q [B, Sq, Hq, D] or grouped [B, Sq, Hkv, G, D]; k/v [B, Sk, Hkv, D] — *views* (any batch / row /
    head / group stride, D contiguous).  Returns o [B, Sq, Hq, D] (contiguous unless `out` given).

    Chunk geometry (zig-zag context parallelism): Sq = len(q_chunk_gid) * chunk_len local rows,
    kv chunk j starts at row kv_chunk_row[j] of k/v.  Defaults: one chunk, gid 0.
    lse_out: fp32 [B, Hq, Sq] contiguous destination of the row log-sum-exp (implies return_lse)."""
    if q.dim() == 5:
        B, Sq, Hkv_q, G, D = q.shape
        Hq = Hkv_q * G
        q_bs, q_rs, q_gs, q_hs = q.stride(0), q.stride(1), q.stride(2), q.stride(3)
    elif q.dim() == 4:
        B, Sq, Hq, D = q.shape
        q_bs, q_rs, q_hs, q_gs = q.stride(0), q.stride(1), q.stride(2), 0
    else:
        raise ValueError("q must be [B, S, H, D] or [B, S, Hkv, G, D]")
    if k.dim() != 4 or v.dim() != 4:
        raise ValueError("k, v must be [B, S, Hkv, D]")
    _, Sk, Hkv, _ = k.shape
    if q.stride(-1) != 1 or k.stride(3) != 1 or v.stride(3) != 1:
        raise ValueError("head_dim must be contiguous")
    if chunk_len is None:
        if Sq != Sk and causal:
            raise ValueError("causal attention without chunk geometry needs Sq == Sk")
        chunk_len = max(Sq, Sk)
        qg, kg, kr = [0], [0], [0]
        q_valid, kv_valid = Sq, Sk
    else:
        qg, kg, kr = list(q_chunk_gid), list(kv_chunk_gid), list(kv_chunk_row)
        if Sq != len(qg) * chunk_len:
            raise ValueError("Sq must equal len(q_chunk_gid) * chunk_len")
        q_valid = kv_valid = chunk_len
    o = torch.empty((B, Sq, Hq, D), dtype=BF16, device=q.device) if out is None else out
    if lse_out is not None:
        if tuple(lse_out.shape) != (B, Hq, Sq) or not lse_out.is_contiguous() or lse_out.dtype != torch.float32:
            raise ValueError("lse_out must be contiguous fp32 [B, Hq, Sq]")
        lse, return_lse = lse_out, True
    else:
        lse = torch.empty((B, Hq, Sq), dtype=torch.float32, device=q.device) if return_lse else None
    p = AttnParams()
    p.q, p.q_batch_stride, p.q_row_stride, p.q_head_stride, p.q_group_stride = _dev(q, "q", BF16), q_bs, q_rs, q_hs, q_gs
    p.k, p.k_batch_stride, p.k_row_stride, p.k_head_stride = _dev(k, "k", BF16), k.stride(0), k.stride(1), k.stride(2)
    p.v, p.v_batch_stride, p.v_row_stride, p.v_head_stride = _dev(v, "v", BF16), v.stride(0), v.stride(1), v.stride(2)
    p.o, p.o_batch_stride, p.o_row_stride, p.o_head_stride, p.o_group_stride = (
        _dev(o, "out", BF16), o.stride(0), o.stride(1), o.stride(2), 0)
    p.lse = _opt(lse, "lse")
    p.batch, p.n_q_heads, p.n_kv_heads, p.head_dim = B, Hq, Hkv, D
    p.chunk_len, p.q_valid, p.kv_valid = chunk_len, q_valid, kv_valid
    p.n_q_chunks, p.n_kv_chunks = len(qg), len(kg)
    qg_a = (C.c_int32 * len(qg))(*qg)
    kg_a = (C.c_int32 * len(kg))(*kg)
    kr_a = (C.c_int64 * len(kr))(*kr)
    p.q_chunk_gid, p.kv_chunk_gid, p.kv_chunk_row = qg_a, kg_a, kr_a
    p.causal = int(causal)
    p.softmax_scale = float(softmax_scale if softmax_scale is not None else 1.0 / math.sqrt(D))
    if seg_start is not None:           # packed sequences: int32 [rows], first row of each query row's segment
        p.q_seg_start = _dev(seg_start, "seg_start", torch.int32)
    _L.check(_L.load().vita_flash_attn_fwd(C.byref(p), _stream()), "vita_flash_attn_fwd")
    return (o, lse) if return_lse else o


def attn_merge_(o_a: torch.Tensor, lse_a: torch.Tensor, o_b: torch.Tensor, lse_b: torch.Tensor) -> torch.Tensor:
    """In place: o_a, lse_a <- the attention over the union of the two DISJOINT key sets the partials (o_a, lse_a) and (o_b, lse_b) saw.
    o [1, S, H, 128] bf16 views (any row / head stride), lse [1, H, S] fp32 contiguous (natural log, -inf = no key seen)."""
    _, S, H, D = o_a.shape
    if o_b.shape != o_a.shape or lse_a.shape != (1, H, S) or lse_b.shape != (1, H, S):
        raise ValueError("attn_merge_: o [1, S, H, D] and lse [1, H, S] of both parts must agree")
    if not (lse_a.is_contiguous() and lse_b.is_contiguous()) or o_a.stride(3) != 1 or o_b.stride(3) != 1:
        raise ValueError("attn_merge_: contiguous lse, unit inner stride of o")
    _L.check(_L.load().vita_attn_merge(_dev(o_a, "o_a", BF16), o_a.stride(1), o_a.stride(2), _dev(lse_a, "lse_a", torch.float32),
                                       _dev(o_b, "o_b", BF16), o_b.stride(1), o_b.stride(2), _dev(lse_b, "lse_b", torch.float32),
                                       S, H, D, _stream()), "vita_attn_merge")
    return o_a


# ------------------------------------------------------------------------------------------------
# ViT helpers
# ------------------------------------------------------------------------------------------------
def patchify14(images: torch.Tensor, k_pad: int = 640, token_major: bool = False) -> torch.Tensor:
    """im2col of Conv2d(3, h, 14, stride 14): [n, 3, H, W] -> [n * P, k_pad] (columns 588.. zero).  token_major: row = patch * n + image
    (Megatron's [s, b, h] order) instead of image * P + patch."""
    n, c, H, W = images.shape
    if c != 3:
        raise ValueError("images must be [n, 3, H, W]")
    im = images.contiguous()
    out = torch.empty((n * (H // 14) * (W // 14), k_pad), dtype=BF16, device=im.device)
    _L.check(_L.load().vita_patchify14_ex(_dev(im, "images", BF16), _dev(out, "patches"), n, H, W, k_pad, int(token_major), _stream()),
             "vita_patchify14")
    return out


def vit_assemble(patch_embeds: torch.Tensor, cls_token: Optional[torch.Tensor], pos_emb: torch.Tensor, n: int,
                 n_patches: int, pos_row0: int = 0, token_major: bool = False) -> torch.Tensor:
    """cat(cls, patch_embeds) + pos[pos_row0 + s] -> [n, seq, h], or [seq, n, h] with token_major (patch_embeds rows token-major too)."""
    hidden = patch_embeds.shape[-1]
    has_cls = cls_token is not None
    seq = n_patches + int(has_cls)
    if pos_emb.shape[0] < pos_row0 + seq:
        raise ValueError("position table too short")
    x = torch.empty((seq, n, hidden) if token_major else (n, seq, hidden), dtype=BF16, device=patch_embeds.device)
    _L.check(_L.load().vita_vit_assemble_ex(_dev(patch_embeds.contiguous(), "patch_embeds", BF16),
                                            _opt(cls_token, "cls_token", BF16), _dev(pos_emb.contiguous(), "pos_emb", BF16),
                                            _dev(x, "x"), n, n_patches, hidden, int(has_cls), int(pos_row0), int(token_major), _stream()),
             "vita_vit_assemble")
    return x


def vit_assemble_bwd(dx: torch.Tensor, token_major: bool = False) -> torch.Tensor:
    """dx [n, seq, h] (or [seq, n, h], token_major) -> bf16 [seq, h] = sum over the images (fp32 accumulation): the gradient of the
    position rows that were looked up; row 0 is also the class token's."""
    if dx.dim() != 3 or not dx.is_contiguous():
        raise ValueError("dx must be a contiguous 3-D tensor")
    seq, n = (dx.shape[0], dx.shape[1]) if token_major else (dx.shape[1], dx.shape[0])
    hidden = dx.shape[2]
    out = torch.empty((seq, hidden), dtype=BF16, device=dx.device)
    if n == 0:
        return out.zero_()
    _L.check(_L.load().vita_vit_assemble_bwd(_dev(dx, "dx", BF16), _dev(out, "d_pos"), n, seq, hidden, int(token_major), _stream()),
             "vita_vit_assemble_bwd")
    return out


def _img_tok_strides(x: torch.Tensor):
    """x [n, seq, h] with unit inner stride (contiguous, or the permuted view of an [s, b, h] tensor): (image, token) strides."""
    if x.dim() != 3 or x.stride(2) != 1:
        raise ValueError("x must be [n, seq, h] with a unit inner stride")
    return x.stride(0), x.stride(1)


def pixel_shuffle_ln(x: torch.Tensor, weight: Optional[torch.Tensor], bias: Optional[torch.Tensor], grid: int, has_cls: bool,
                     eps: float = 1e-5, norm: bool = True) -> torch.Tensor:
    """drop cls + pixel_shuffle(0.5) (+ LayerNorm(4 h) when norm): x [n, seq, h] (any image / token stride) -> [n, (grid/2)^2, 4 h]."""
    n, _, hidden = x.shape
    if x.stride(2) != 1 or (x.stride(0) | x.stride(1)) & 7:
        x = x.contiguous()
    si, st = _img_tok_strides(x)
    y = torch.empty((n, (grid // 2) ** 2, hidden * 4), dtype=BF16, device=x.device)
    if n == 0:
        return y
    _L.check(_L.load().vita_pixel_shuffle_ln_ex(_dev(x, "x", BF16), _opt(weight, "weight", BF16), _opt(bias, "bias", BF16),
                                                _dev(y, "y"), n, grid, hidden, int(has_cls), float(eps), int(norm), si, st, _stream()),
             "vita_pixel_shuffle_ln")
    return y


def pixel_shuffle_ln_bwd(dy: torch.Tensor, x: torch.Tensor, weight: Optional[torch.Tensor], grid: int, has_cls: bool, eps: float,
                         dgamma: Optional[torch.Tensor], dbeta: Optional[torch.Tensor], want_dx: bool = True, norm: bool = True):
    """Backward of pixel_shuffle_ln: returns dx (x's shape, contiguous; the class token's row zero) or None; dgamma / dbeta fp32 [4 h]
    (zeroed by the caller) accumulate the LayerNorm parameter gradients."""
    n, seq, hidden = x.shape
    if x.stride(2) != 1 or (x.stride(0) | x.stride(1)) & 7:
        x = x.contiguous()
    dx = torch.empty((n, seq, hidden), dtype=BF16, device=x.device) if want_dx else None
    if n == 0:
        return dx
    if want_dx and not x.is_contiguous():
        # dx is written with x's strides: give it the same layout ([s, b, h] memory, viewed [n, seq, h])
        dx = torch.empty_strided((n, seq, hidden), x.stride(), dtype=BF16, device=x.device)
    si, st = _img_tok_strides(x)
    _L.check(_L.load().vita_pixel_shuffle_ln_bwd(_dev(dy.contiguous(), "dy", BF16), _dev(x, "x", BF16), _opt(weight, "weight", BF16),
                                                 _opt(dx, "dx", BF16), _opt(dgamma, "dgamma", torch.float32),
                                                 _opt(dbeta, "dbeta", torch.float32), n, grid, hidden, int(has_cls), float(eps),
                                                 int(norm), si, st, _stream()), "vita_pixel_shuffle_ln_bwd")
    return dx


# ------------------------------------------------------------------------------------------------
# backward
# ------------------------------------------------------------------------------------------------
def transpose(x: torch.Tensor, out: Optional[torch.Tensor] = None) -> torch.Tensor:
    """x [R, C] (row stride allowed) -> [C, R] contiguous."""
    if x.dim() != 2 or x.stride(1) != 1:
        raise ValueError("x must be 2-D with contiguous last dim")
    R, Cc = x.shape
    y = torch.empty((Cc, R), dtype=BF16, device=x.device) if out is None else out
    _L.check(_L.load().vita_transpose_bf16(_dev(x, "x", BF16), x.stride(0), _dev(y, "out", BF16), y.stride(0), R, Cc,
                                           _stream()), "vita_transpose_bf16")
    return y


def rope_qkv_bwd_(d_mixed: torch.Tensor, groups: int, q_per_group: int, head_dim: int, cos, sin) -> torch.Tensor:
    rows = d_mixed.numel() // (groups * (q_per_group + 2) * head_dim)
    _L.check(_L.load().vita_rope_qkv_bwd(_dev(d_mixed, "d_mixed", BF16), rows, groups, q_per_group, head_dim,
                                         _dev(cos, "cos", BF16), _dev(sin, "sin", BF16), _stream()), "vita_rope_qkv_bwd")
    return d_mixed


def rmsnorm_bwd(dy, x, weight, eps: float, dw_acc: Optional[torch.Tensor] = None, out=None,
                residual: Optional[torch.Tensor] = None) -> torch.Tensor:
    cols = x.shape[-1]
    rows = x.numel() // cols
    dx = torch.empty_like(x) if out is None else out
    if rows == 0:
        return dx
    _L.check(_L.load().vita_rmsnorm_bwd(_dev(dy, "dy", BF16), _dev(x, "x", BF16), _dev(weight, "weight", BF16),
                                        _opt(residual, "residual", BF16), _dev(dx, "dx", BF16), _opt(dw_acc, "dw_acc", torch.float32), rows, cols,
                                        float(eps), _stream()), "vita_rmsnorm_bwd")
    return dx


def swiglu(y: torch.Tensor, out=None) -> torch.Tensor:
    rows, two_f = y.shape
    a = torch.empty((rows, two_f // 2), dtype=BF16, device=y.device) if out is None else out
    if rows == 0:
        return a
    _L.check(_L.load().vita_swiglu_fwd(_dev(y, "y", BF16), _dev(a, "a", BF16), rows, two_f // 2, _stream()),
             "vita_swiglu_fwd")
    return a


def swiglu_bwd(y: torch.Tensor, da: torch.Tensor, out=None) -> torch.Tensor:
    rows, two_f = y.shape
    dy = torch.empty_like(y) if out is None else out
    if rows == 0:
        return dy
    _L.check(_L.load().vita_swiglu_bwd(_dev(y, "y", BF16), _dev(da, "da", BF16), _dev(dy, "dy", BF16), rows,
                                       two_f // 2, _stream()), "vita_swiglu_bwd")
    return dy


def gelu_bwd(x: torch.Tensor, dy: torch.Tensor, tanh: bool = False) -> torch.Tensor:
    dx = torch.empty_like(x)
    if x.numel() == 0:
        return dx
    fn = _L.load().vita_gelu_tanh_bwd if tanh else _L.load().vita_gelu_bwd
    _L.check(fn(_dev(x, "x", BF16), _dev(dy, "dy", BF16), _dev(dx, "dx", BF16), x.numel(), _stream()),
             "vita_gelu_tanh_bwd" if tanh else "vita_gelu_bwd")
    return dx


def layernorm_bwd(dy: torch.Tensor, x: torch.Tensor, weight: torch.Tensor, eps: float, dgamma: torch.Tensor, dbeta: torch.Tensor,
                  out: Optional[torch.Tensor] = None) -> torch.Tensor:
    """Backward of layernorm(): returns dx; dgamma / dbeta (fp32 [cols], zeroed by the caller) receive the parameter gradients."""
    cols = x.shape[-1]
    dx = torch.empty_like(x) if out is None else out
    if x.numel() == 0:                      # an empty frame batch: no rows, dgamma / dbeta stay as the caller zeroed them
        return dx
    _L.check(_L.load().vita_layernorm_bwd(_dev(dy, "dy", BF16), _dev(x, "x", BF16), _dev(weight, "weight", BF16), _dev(dx, "dx", BF16),
                                          _dev(dgamma, "dgamma", torch.float32), _dev(dbeta, "dbeta", torch.float32),
                                          x.numel() // cols, cols, float(eps), _stream()), "vita_layernorm_bwd")
    return dx


def gelu(x: torch.Tensor, tanh: bool = False) -> torch.Tensor:
    a = torch.empty_like(x)
    if x.numel() == 0:
        return a
    _L.check(_L.load().vita_gelu_fwd(_dev(x, "x", BF16), _dev(a, "a", BF16), x.numel(), int(tanh), _stream()), "vita_gelu_fwd")
    return a


def bias_scale_residual(x: torch.Tensor, bias: Optional[torch.Tensor], scale: Optional[torch.Tensor], residual: torch.Tensor) -> torch.Tensor:
    """bf16(residual + bf16(bf16(x + bias) * scale)) — InternViTTransformerLayer's LayerScale residual; bias / scale optional."""
    cols = x.shape[-1]
    out = torch.empty_like(x)
    if x.numel() == 0:
        return out
    _L.check(_L.load().vita_bias_scale_res_fwd(_dev(x, "x", BF16), _opt(bias, "bias", BF16), _opt(scale, "scale", BF16),
                                               _dev(residual, "residual", BF16), _dev(out, "out", BF16), x.numel() // cols, cols,
                                               _stream()), "vita_bias_scale_res_fwd")
    return out


def bias_scale_residual_bwd(g: torch.Tensor, x: torch.Tensor, bias, scale, d_bias: Optional[torch.Tensor],
                            d_scale: Optional[torch.Tensor]) -> torch.Tensor:
    """-> dx = bf16(g * scale) (g itself when scale is None); d_bias / d_scale fp32 [cols] accumulate (zeroed by the caller)."""
    cols = x.shape[-1]
    dx = torch.empty_like(x) if scale is not None else None
    if x.numel() == 0:
        return g if dx is None else dx
    _L.check(_L.load().vita_bias_scale_res_bwd(_dev(g, "g", BF16), _dev(x, "x", BF16), _opt(bias, "bias", BF16), _opt(scale, "scale", BF16),
                                               _opt(dx, "dx", BF16), _opt(d_bias, "d_bias", torch.float32),
                                               _opt(d_scale, "d_scale", torch.float32), x.numel() // cols, cols, _stream()),
             "vita_bias_scale_res_bwd")
    return g if dx is None else dx


def layernorm_param_grad(dy, x, dgamma: torch.Tensor, dbeta: torch.Tensor, eps: float,
                         prenormalized: bool = False) -> None:
    cols = x.shape[-1]
    _L.check(_L.load().vita_layernorm_param_grad(_dev(dy, "dy", BF16), _dev(x, "x", BF16),
                                                 _dev(dgamma, "dgamma", torch.float32),
                                                 _dev(dbeta, "dbeta", torch.float32), x.numel() // cols, cols,
                                                 float(eps), int(prenormalized), _stream()),
             "vita_layernorm_param_grad")


def ce_loss(logits: torch.Tensor, labels: torch.Tensor, grad_scale: Optional[torch.Tensor] = None,
            want_grad: bool = False, want_loss: bool = True, strict: bool = True):
    """logits [n, V] bf16 (or fp32: dlogits fp32 too), labels [n] int64 -> loss [n] fp32 (, dlogits [n, V]); grad_scale fp32 [n].
    A label outside [0, V): the kernel follows Megatron's masked-target rule (loss = log sum exp(l - max), no one-hot term in the
    gradient; the datasets pad with -100 and the loss mask removes those rows).  strict=True (the stand-alone step, whose labels are
    always real tokens) additionally raises IndexError for such a label; strict=False neither checks nor synchronises."""
    n, V = logits.shape
    f32 = logits.dtype == torch.float32
    if not f32 and logits.dtype != BF16:
        raise ValueError("logits must be bf16 or fp32")
    if logits.stride(1) != 1:
        logits = logits.contiguous()
    loss = torch.empty(n, dtype=torch.float32, device=logits.device) if (want_loss or not f32) else None
    dl = torch.empty_like(logits, memory_format=torch.contiguous_format) if want_grad else None
    if n == 0:
        return (loss, dl) if want_grad else loss
    flag = _err_flag(logits.device) if strict else None
    fn = _L.load().vita_ce_loss_f32 if f32 else _L.load().vita_ce_loss
    _L.check(fn(_dev(logits, "logits"), logits.stride(0), _dev(labels.contiguous(), "labels", torch.int64), _opt(loss, "loss"),
                _opt(dl, "dlogits"), dl.stride(0) if want_grad else 0, _opt(grad_scale, "grad_scale", torch.float32), n, V,
                _opt(flag, "flag"), _stream()), "vita_ce_loss")
    if strict and int(flag.item()):
        raise IndexError("vita_ce_loss: label out of range")
    return (loss, dl) if want_grad else loss


def _ce_vp_logits(logits: torch.Tensor):
    if logits.dtype not in (torch.float32, BF16):
        raise ValueError("logits must be bf16 or fp32")
    if logits.dim() != 2:
        raise ValueError("logits must be [rows, vocab_local]")
    return logits if logits.stride(1) == 1 else logits.contiguous()


def ce_vp_stats(logits: torch.Tensor, labels: torch.Tensor, vocab_start: int) -> torch.Tensor:
    """Vocabulary-parallel cross entropy, pass 1 on this rank's shard [n, V / TP] (labels in global ids): fp32 [n, 4] =
    {max, sum exp(l - max), predicted raw logit or 0, label-in-shard 0 / 1} — the record the tensor-parallel group all-gathers."""
    logits = _ce_vp_logits(logits)
    n, v_l = logits.shape
    stats = torch.empty((n, 4), dtype=torch.float32, device=logits.device)
    _L.check(_L.load().vita_ce_vp_stats(_dev(logits, "logits"), int(logits.dtype == torch.float32), logits.stride(0),
                                        _dev(labels.contiguous(), "labels", torch.int64), int(vocab_start), _dev(stats, "stats"), n, v_l,
                                        _stream()), "vita_ce_vp_stats")
    return stats


def ce_vp_finish(stats_all: torch.Tensor, want_loss: bool = True):
    """stats_all fp32 [tp, n, 4] (rank-major) -> (loss [n] fp32 or None, row_stat [n, 2] = {global max, global sum exp})."""
    tp, n, four = stats_all.shape
    if four != 4 or not stats_all.is_contiguous():
        raise ValueError("stats_all must be contiguous [tp, rows, 4]")
    loss = torch.empty(n, dtype=torch.float32, device=stats_all.device) if want_loss else None
    row_stat = torch.empty((n, 2), dtype=torch.float32, device=stats_all.device)
    _L.check(_L.load().vita_ce_vp_finish(_dev(stats_all, "stats_all", torch.float32), tp, n, _opt(loss, "loss"), _dev(row_stat, "row_stat"),
                                         _stream()), "vita_ce_vp_finish")
    return loss, row_stat


def ce_vp_grad(logits: torch.Tensor, labels: torch.Tensor, vocab_start: int, row_stat: torch.Tensor,
               grad_scale: Optional[torch.Tensor] = None, out: Optional[torch.Tensor] = None) -> torch.Tensor:
    """dlogits of this rank's shard only: (exp(l - max) / sumexp - onehot(label - vocab_start)) * grad_scale.  `out` may be the
    logits tensor itself (element-wise, in place — Megatron reuses the buffer the same way)."""
    logits = _ce_vp_logits(logits)
    n, v_l = logits.shape
    dl = torch.empty_like(logits) if out is None else out
    if dl.shape != logits.shape or dl.dtype != logits.dtype or dl.stride(1) != 1:
        raise ValueError("out must match the logits' shape / dtype with unit column stride")
    _L.check(_L.load().vita_ce_vp_grad(_dev(logits, "logits"), int(logits.dtype == torch.float32), logits.stride(0),
                                       _dev(labels.contiguous(), "labels", torch.int64), int(vocab_start),
                                       _dev(row_stat, "row_stat", torch.float32), _opt(grad_scale, "grad_scale", torch.float32),
                                       _dev(dl, "dlogits"), dl.stride(0), n, v_l, _stream()), "vita_ce_vp_grad")
    return dl


def row_scatter_add_f32_(dst: torch.Tensor, idx: torch.Tensor, src: torch.Tensor) -> torch.Tensor:
    n, cols = src.shape
    if n == 0:
        return dst
    flag = _err_flag(dst.device)
    _L.check(_L.load().vita_row_scatter_add_f32(_dev(src, "src", BF16), _dev(idx.contiguous(), "idx", torch.int64),
                                                _dev(dst, "dst", torch.float32), dst.shape[0], n, cols,
                                                _dev(flag, "flag"), _stream()), "vita_row_scatter_add_f32")
    if int(flag.item()):
        raise IndexError("vita_row_scatter_add_f32: index out of range")
    return dst


ATTN_BWD_DQ, ATTN_BWD_DKV = 1, 2


def attn_delta(o: torch.Tensor, d_o: torch.Tensor, delta: torch.Tensor) -> torch.Tensor:
    """delta[h, row] = sum_d float(dO[row, h, d]) * float(O[row, h, d]) — the backward's pre-pass; o, d_o [1, Sq, Hq, D] views, delta fp32 [Hq, Sq]."""
    _, Sq, Hq, D = o.shape
    _L.check(_L.load().vita_attn_delta(_dev(o, "o", BF16), _dev(d_o, "d_o", BF16), _dev(delta, "delta", torch.float32), Sq, Hq, D,
                                       o.stride(1), o.stride(2), d_o.stride(1), d_o.stride(2), _stream()), "vita_attn_delta")
    return delta


def flash_attn_bwd(q5, k, v, o, d_o, lse, *, chunk_len=None, q_chunk_gid=None, kv_chunk_gid=None, kv_chunk_row=None,
                   softmax_scale=None, dq5=None, dk=None, dv=None, seg_start=None, seg_end=None, parts=ATTN_BWD_DQ | ATTN_BWD_DKV,
                   delta=None):
    """Backward of flash_attn(causal=True) at batch 1, head_dim 128 (seg_start / seg_end int32 [rows]: packed sequences).
    q5 [1, Sq, Hkv, G, D] (or [1, Sq, Hq, D]); k, v [1, Sk, Hkv, D] views; o, d_o [1, Sq, Hq, D];
    lse [1, Hq, Sq].  Returns (dq like q5, dk, dv like k/v — dk/dv cover every row of k/v).
    parts: ATTN_BWD_DKV / ATTN_BWD_DQ run one pass only (a context-parallel caller starts the dK / dV reduce-scatter between them);
    delta: the fp32 [Hq, Sq] row sums of dO * O from an earlier call of this function (returned as 4th value when parts is partial)."""
    if q5.dim() == 5:
        _, Sq, Hkv_q, G, D = q5.shape
        Hq = Hkv_q * G
        q_rs, q_gs, q_hs = q5.stride(1), q5.stride(2), q5.stride(3)
    else:
        _, Sq, Hq, D = q5.shape
        q_rs, q_hs, q_gs = q5.stride(1), q5.stride(2), 0
    _, Sk, Hkv, _ = k.shape
    if chunk_len is None:
        chunk_len, qg, kg, kr = Sq, [0], [0], [0]
    else:
        qg, kg, kr = list(q_chunk_gid), list(kv_chunk_gid), list(kv_chunk_row)
    both = parts == (ATTN_BWD_DQ | ATTN_BWD_DKV)
    if parts & ATTN_BWD_DQ:
        dq5 = torch.empty_like(q5) if dq5 is None else dq5
    if parts & ATTN_BWD_DKV:
        dk = torch.empty_like(k) if dk is None else dk
        dv = torch.empty_like(v) if dv is None else dv
    h = _L.load()
    if delta is None:
        delta = torch.empty((Hq, Sq), dtype=torch.float32, device=q5.device)
        _L.check(h.vita_attn_delta(_dev(o, "o", BF16), _dev(d_o, "d_o", BF16), _dev(delta, "delta"), Sq, Hq, D,
                                   o.stride(1), o.stride(2), d_o.stride(1), d_o.stride(2), _stream()), "vita_attn_delta")
    p = _L.AttnBwdParams()
    p.q, p.q_row_stride, p.q_head_stride, p.q_group_stride = _dev(q5, "q", BF16), q_rs, q_hs, q_gs
    p.k, p.k_row_stride, p.k_head_stride = _dev(k, "k", BF16), k.stride(1), k.stride(2)
    p.v, p.v_row_stride, p.v_head_stride = _dev(v, "v", BF16), v.stride(1), v.stride(2)
    p.d_o, p.do_row_stride, p.do_head_stride = _dev(d_o, "d_o", BF16), d_o.stride(1), d_o.stride(2)
    p.lse, p.delta = _dev(lse, "lse", torch.float32), _dev(delta, "delta")
    if dq5 is not None:
        if dq5.dim() == 5:
            p.dq, p.dq_row_stride, p.dq_group_stride, p.dq_head_stride = (_dev(dq5, "dq", BF16), dq5.stride(1),
                                                                           dq5.stride(2), dq5.stride(3))
        else:
            p.dq, p.dq_row_stride, p.dq_head_stride, p.dq_group_stride = _dev(dq5, "dq", BF16), dq5.stride(1), dq5.stride(2), 0
    if dk is not None:
        p.dk, p.dk_row_stride, p.dk_head_stride = _dev(dk, "dk", BF16), dk.stride(1), dk.stride(2)
        p.dv, p.dv_row_stride, p.dv_head_stride = _dev(dv, "dv", BF16), dv.stride(1), dv.stride(2)
    p.n_q_heads, p.n_kv_heads, p.head_dim = Hq, Hkv, D
    p.chunk_len, p.n_q_chunks, p.n_kv_chunks = chunk_len, len(qg), len(kg)
    qg_a, kg_a, kr_a = (C.c_int32 * len(qg))(*qg), (C.c_int32 * len(kg))(*kg), (C.c_int64 * len(kr))(*kr)
    p.q_chunk_gid, p.kv_chunk_gid, p.kv_chunk_row = qg_a, kg_a, kr_a
    p.softmax_scale = float(softmax_scale if softmax_scale is not None else 1.0 / math.sqrt(D))
    if seg_start is not None:
        p.q_seg_start, p.k_seg_end = _dev(seg_start, "seg_start", torch.int32), _dev(seg_end, "seg_end", torch.int32)
    _L.check(h.vita_flash_attn_bwd_parts(C.byref(p), int(parts), _stream()), "vita_flash_attn_bwd")
    return (dq5, dk, dv) if both else (dq5, dk, dv, delta)


# ------------------------------------------------------------------------------------------------
# single-token decode against the sharded KV cache
# ------------------------------------------------------------------------------------------------
DECODE_KEYS_PER_TILE = 128
DECODE_MAX_SPLITS = 128


def gemv(x: torch.Tensor, w: torch.Tensor, epilogue: int = EPI_NONE, bias: Optional[torch.Tensor] = None,
         residual: Optional[torch.Tensor] = None, out: Optional[torch.Tensor] = None) -> torch.Tensor:
    """y[N] = epilogue(W[N,K] . x[K]) for one token; SWIGLU: W = cat[gate, up] rows, N = W.shape[0] / 2."""
    K = x.numel()
    n_rows = w.shape[0]
    if w.shape[1] != K:
        raise RuntimeError(f"supplied weight's shape is {tuple(w.shape)}, K = {K} expected")
    N = n_rows // 2 if epilogue == EPI_SWIGLU else n_rows
    y = torch.empty(N, dtype=BF16, device=x.device) if out is None else out
    if y.numel() != N or not y.is_contiguous() or not x.is_contiguous():
        raise ValueError("gemv needs contiguous x and out of N elements")
    _L.check(_L.load().vita_gemv_bf16(_dev(x, "x", BF16), _dev(w, "w", BF16), w.stride(0), _dev(y, "out", BF16), N, K,
                                      epilogue, _opt(bias, "bias", BF16), _opt(residual, "residual", BF16), _stream()),
             "vita_gemv_bf16")
    return y


def decode_splits(length: int) -> int:
    return max(1, min(DECODE_MAX_SPLITS, -(-length // DECODE_KEYS_PER_TILE)))


def decode_attn_partial(q: torch.Tensor, k_cache: torch.Tensor, v_cache: torch.Tensor, length: int,
                        softmax_scale: Optional[float] = None, len_dev: Optional[torch.Tensor] = None):
    """q [groups, qpg, d] (any strides with unit inner stride), caches [cap, groups, d] views; attends to rows
    [0, length) — or [0, min(length, len_dev[0])) with the count read on the device (int32 [1]; hipGraph replay:
    pass length = capacity).  Returns the un-normalised partials (m [n, H], l [n, H], o [n, H, d]) fp32, log2
    domain."""
    G, qpg, d = q.shape
    if k_cache.shape[1:] != (G, d) or v_cache.shape != k_cache.shape or q.stride(2) != 1 or k_cache.stride(2) != 1:
        raise ValueError("decode_attn_partial: q [G, qpg, d], caches [cap, G, d]")
    if k_cache.stride() != v_cache.stride() or length > k_cache.shape[0]:
        raise ValueError("k/v cache views must share strides and hold `length` rows")
    H = G * qpg
    n = DECODE_MAX_SPLITS if len_dev is not None else decode_splits(length)
    pm = torch.empty(n, H, dtype=torch.float32, device=q.device)
    pl = torch.empty(n, H, dtype=torch.float32, device=q.device)
    po = torch.empty(n, H, d, dtype=torch.float32, device=q.device)
    if length == 0:
        pm.fill_(float("-inf")); pl.zero_(); po.zero_()
        return pm, pl, po
    scale = 1.0 / math.sqrt(d) if softmax_scale is None else softmax_scale
    _L.check(_L.load().vita_decode_attn_partial(_dev(q, "q", BF16), q.stride(0), q.stride(1), _dev(k_cache, "k", BF16),
                                                _dev(v_cache, "v", BF16), k_cache.stride(0), k_cache.stride(1),
                                                int(length), _opt(len_dev, "len_dev", torch.int32), n, G, qpg, d,
                                                float(scale), _dev(pm, "pm"), _dev(pl, "pl"), _dev(po, "po"),
                                                _stream()), "vita_decode_attn_partial")
    return pm, pl, po


def decode_attn_merge(pm: torch.Tensor, pl: torch.Tensor, po: torch.Tensor, final: bool,
                      out: Optional[torch.Tensor] = None, packed_out: Optional[torch.Tensor] = None):
    """Merge partials over dim 0 (parts may be strided on dim 0).  final: bf16 context [H, d]; else one merged
    partial written into packed_out (fp32 [H*d + 2H]: o, then m, then l) — the message the CP ranks exchange."""
    n, H, d = po.shape
    for t_ in (pm, pl, po):
        if t_.dtype != torch.float32 or t_.stride(-1) != 1:
            raise ValueError("partials must be fp32 with unit inner stride")
    if pm.stride(0) != pl.stride(0) or (n > 0 and po.stride(1) != d):
        raise ValueError("partials: m and l must share the part stride, o rows must be dense")
    L = _L.load()
    if final:
        ctx = torch.empty(H, d, dtype=BF16, device=po.device) if out is None else out
        _L.check(L.vita_decode_attn_merge(_dev(pm, "pm"), _dev(pl, "pl"), _dev(po, "po"), n, pm.stride(0), po.stride(0),
                                          H, d, None, None, None, _dev(ctx, "out", BF16), _stream()),
                 "vita_decode_attn_merge")
        return ctx
    buf = torch.empty(H * d + 2 * H, dtype=torch.float32, device=po.device) if packed_out is None else packed_out
    base = _dev(buf, "packed_out", torch.float32)
    _L.check(L.vita_decode_attn_merge(_dev(pm, "pm"), _dev(pl, "pl"), _dev(po, "po"), n, pm.stride(0), po.stride(0),
                                      H, d, base + 4 * H * d, base + 4 * (H * d + H), base, None, _stream()),
             "vita_decode_attn_merge")
    return buf


def unpack_partials(gathered: torch.Tensor, heads: int, d: int):
    """gathered [n, H*d + 2H] fp32 (decode_attn_merge packed messages) -> strided (m, l, o) views."""
    n = gathered.shape[0]
    o = gathered[:, : heads * d].view(n, heads, d)
    return gathered[:, heads * d: heads * d + heads], gathered[:, heads * d + heads:], o


def segments_from_cu_seqlens(cu_seqlens: torch.Tensor, total: int):
    """cu_seqlens [n + 1] (PackedSeqParams.cu_seqlens_q, M/training/utils.py:31-57) -> (seg_start, seg_end) int32 [total]:
    for every row the first row and one-past-the-last row of its packed sample (rows past cu_seqlens[-1] form a last
    segment of their own, as the reference appends seq_length to actual_seq_len).  Index plumbing, plain torch."""
    cu = cu_seqlens.to(torch.int64).reshape(-1)
    if int(cu[-1]) < total:
        cu = torch.cat([cu, cu.new_tensor([total])])
    lens = cu[1:] - cu[:-1]
    start = torch.repeat_interleave(cu[:-1], lens).to(torch.int32)
    end = torch.repeat_interleave(cu[1:], lens).to(torch.int32)
    return start.contiguous(), end.contiguous()


def logit_postprocess_(logits: torch.Tensor, multiplier_scale: float = 0.0, softcapping: float = 0.0) -> torch.Tensor:
    """In place on bf16 logits [..., V]: x * scale, then tanh(x / cap) * cap (gpt_vl_model.py:349-355); 0 / None = skip."""
    s, c = float(multiplier_scale or 0.0), float(softcapping or 0.0)
    if s == 0.0 and c == 0.0:
        return logits
    x = logits.view(-1, logits.shape[-1])
    _L.check(_L.load().vita_logit_postprocess(_dev(x, "logits", BF16), x.stride(0), x.shape[0], x.shape[1], s, c, _stream()),
             "vita_logit_postprocess")
    return logits


def logit_postprocess_bwd_(y: torch.Tensor, grad: torch.Tensor, multiplier_scale: float = 0.0, softcapping: float = 0.0) -> torch.Tensor:
    """grad (bf16, in place) *= d(post-processed logits) / d(raw logits), from the post-processed logits y."""
    s, c = float(multiplier_scale or 0.0), float(softcapping or 0.0)
    if s == 0.0 and c == 0.0:
        return grad
    y2, g2 = y.view(-1, y.shape[-1]), grad.view(-1, grad.shape[-1])
    _L.check(_L.load().vita_logit_postprocess_bwd(_dev(y2, "y", BF16), y2.stride(0), _dev(g2, "grad", BF16), g2.stride(0), g2.shape[0],
                                                  g2.shape[1], s, c, _stream()), "vita_logit_postprocess_bwd")
    return grad


def add(a: torch.Tensor, b: torch.Tensor, out: Optional[torch.Tensor] = None) -> torch.Tensor:
    """out = bf16(a + b) (out may alias a): one pass, no copy of either operand."""
    if a.shape != b.shape or not a.is_contiguous() or not b.is_contiguous():
        raise ValueError("add: contiguous tensors of equal shape")
    y = torch.empty_like(a) if out is None else out
    if a.numel() == 0:
        return y
    _L.check(_L.load().vita_add_bf16(_dev(a, "a", BF16), _dev(b, "b", BF16), _dev(y, "out", BF16), a.numel(), _stream()), "vita_add_bf16")
    return y


def add_(a: torch.Tensor, b: torch.Tensor) -> torch.Tensor:
    """a = bf16(a + b) in place (residual add behind a tensor-parallel all-reduce)."""
    if a.shape != b.shape or not a.is_contiguous() or not b.is_contiguous():
        raise ValueError("add_: contiguous tensors of equal shape")
    _L.check(_L.load().vita_add_bf16(_dev(a, "a", BF16), _dev(b, "b", BF16), _dev(a, "a", BF16), a.numel(), _stream()),
             "vita_add_bf16")
    return a
