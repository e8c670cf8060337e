"""torch.autograd bridges over libvita_hip.so for the Megatron-constructible modules (layers.py,
language_model_embedding.py, dot_product_attention.py, rotary_pos_embedding.py).

Megatron gets its backward from autograd over its modules (`ctx.save_for_backward`,
M/core/tensor_parallel/layers.py:382); a module that replaces one of them therefore has to be an autograd node.
Every forward and backward below is a library kernel — the same calls the explicit sweep of training.TrainStep makes.
"""
from __future__ import annotations

import torch
import torch.distributed as dist

from . import ops, parallel_state as mpu


# ------------------------------------------------------------------------------------------------
# GEMM helpers shared with training.TrainStep (one NT GEMM serves forward, dgrad and wgrad)
# ------------------------------------------------------------------------------------------------
def transpose(x: torch.Tensor) -> torch.Tensor:
    return ops.transpose(x)


def pad_rows(x: torch.Tensor, mult: int = 64) -> torch.Tensor:
    """Zero rows up to a multiple of `mult` (the contraction length of a wgrad GEMM is the row count)."""
    m = x.shape[0]
    if m % mult == 0:
        return x
    out = torch.zeros((m + mult - m % mult,) + tuple(x.shape[1:]), dtype=x.dtype, device=x.device)
    out[:m] = x
    return out


def dgrad(dy: torch.Tensor, w: torch.Tensor, out=None, epilogue=ops.EPI_NONE, residual=None) -> torch.Tensor:
    """grad_input = grad_output.matmul(weight)  (layers.py:444):  dy [M, N], w [N, K] -> [M, K].  Whole-tile shapes (every decoder
    linear at its real sizes) take vita_gemm_bf16_nn: the weight is read contraction-major as it lies; others transpose it first."""
    if epilogue == ops.EPI_NONE and residual is None and ops.gemm_nn_ok(dy, w) and (out is None or out.stride(1) == 1):
        return ops.gemm_nn(dy, w, out=out)
    return ops.gemm(dy, transpose(w), epilogue, residual=residual, out=out)


def wgrad(dy_t: torch.Tensor, x: torch.Tensor) -> torch.Tensor:
    """grad_weight = grad_output.t().matmul(total_input)  (layers.py:522-523): dy_t [N, M], x [M, K] -> [N, K]."""
    return ops.gemm(dy_t, transpose(x))


def bias_grad(dy: torch.Tensor) -> torch.Tensor:
    """grad_bias = grad_output.sum(dim=0) (layers.py:524): dy [rows, N] -> [N], one pass over dy (vita_colsum_bf16, fp32 sums; through
    r03 a GEMM against a block of ones: a whole-sequence contraction on N / 256 workgroups)."""
    if dy.shape[1] % 4 or dy.stride(1) != 1:
        dy = dy.contiguous()
        if dy.shape[1] % 4:
            return dy.float().sum(dim=0).to(dy.dtype)        # odd widths (test-sized layers only)
    return ops.colsum(dy).to(dy.dtype)


def _tp():
    return mpu.get_tensor_model_parallel_world_size(), mpu.get_tensor_model_parallel_group()


def weight_bias_grads(go: torch.Tensor, x: torch.Tensor, need_weight: bool, need_bias: bool, n_out: int, dtype):
    """grad_weight = grad_output.t().matmul(total_input), grad_bias = grad_output.sum(dim=0) (layers.py:522-524) for go [rows, n_out],
    x [rows, k].  The TN kernel takes both operands contraction-major as they are; shapes it does not tile (output dims not multiples
    of 256) go through two vita_transpose_bf16 passes and the NT GEMM.  No rows (an empty logit-mask selection on this rank): zeros."""
    grad_weight = grad_bias = None
    if go.shape[0] == 0:
        if need_weight:
            grad_weight = torch.zeros(n_out, x.shape[1], dtype=dtype, device=go.device)
        if need_bias:
            grad_bias = torch.zeros(n_out, dtype=dtype, device=go.device)
        return grad_weight, grad_bias
    if not (need_weight or need_bias):
        return None, None
    if need_bias:
        grad_bias = bias_grad(go).to(dtype)                                               # :524
    if need_weight:
        go_p, x_p = pad_rows(go), pad_rows(x.contiguous())
        if ops.gemm_tn_ok(go_p, x_p):            # both operands contraction-major as they are: no transposed copies
            grad_weight = ops.gemm_tn(go_p, x_p)                                          # :522-523
        else:
            grad_weight = wgrad(transpose(go_p), x_p)
    return grad_weight, grad_bias


# ------------------------------------------------------------------------------------------------
# LinearWithGradAccumulationAndAsyncCommunication (M/core/tensor_parallel/layers.py:366-534), logit_mask included
# ------------------------------------------------------------------------------------------------
class LinearFn(torch.autograd.Function):
    """input [s, b, c] (sequence-parallel: the rank's s / TP rows), weight [n, c], bias [n] | None ->
    output [s' , b, n] with s' = the logit-masked row count, else the (gathered) sequence length."""

    @staticmethod
    def forward(ctx, input, weight, bias, allreduce_dgrad, sequence_parallel, logit_mask):
        ctx.save_for_backward(input, weight, logit_mask)
        ctx.use_bias = bias is not None
        ctx.allreduce_dgrad, ctx.sequence_parallel = allreduce_dgrad, sequence_parallel
        ctx.tp, ctx.group = _tp()       # captured here: the backward runs on autograd's own thread
        total_input = _gather_sequence(input, ctx.tp, ctx.group) if sequence_parallel else input  # :392-400
        s, b, c = total_input.shape
        x = total_input.reshape(s * b, c)
        if logit_mask is not None:                                                                 # :402-407
            if b != 1:
                raise AssertionError("logit_mask requires batch 1 (gpt_vl_model.py:329)")
            x = ops.row_gather(x.contiguous(), ops.mask_to_index(logit_mask.transpose(0, 1).reshape(-1)))
        m = x.shape[0]
        if m <= 16 and bias is None:
            out = ops.gemm_skinny(x, weight)
        else:
            out = ops.gemm(x, weight, ops.EPI_BIAS if bias is not None else ops.EPI_NONE, bias)   # :409-411
        return out.view(m // b, b, out.shape[-1])     # explicit width: m == 0 (no row selected on this rank) must reshape too

    @staticmethod
    def backward(ctx, grad_output):
        input, weight, logit_mask = ctx.saved_tensors
        tp, group = ctx.tp, ctx.group
        total_input = _gather_sequence(input, tp, group) if ctx.sequence_parallel else input      # :435-452
        s, b, c = total_input.shape
        go = grad_output.reshape(-1, grad_output.shape[-1]).contiguous()
        grad_input = dgrad(go, weight)                                                             # :453
        x = total_input.reshape(s * b, c)
        if logit_mask is not None:                                                                 # :455-466
            idx = ops.mask_to_index(logit_mask.transpose(0, 1).reshape(-1))
            full = torch.zeros(s * b, c, dtype=grad_input.dtype, device=grad_input.device)
            grad_input = ops.row_scatter_(full, idx, grad_input)
            x = ops.row_gather(x.contiguous(), idx)
        grad_input = grad_input.view(s, b, c)
        if ctx.allreduce_dgrad and tp > 1:                                                        # :475-481
            dist.all_reduce(grad_input, group=group)
        if ctx.sequence_parallel:                                                                 # :483-494
            sub = torch.empty(input.shape, dtype=input.dtype, device=input.device)
            dist.reduce_scatter_tensor(sub, grad_input.contiguous(), group=group)
            grad_input = sub
        grad_weight, grad_bias = weight_bias_grads(go, x, ctx.needs_input_grad[1], ctx.use_bias, weight.shape[0], weight.dtype)
        return grad_input, grad_weight, grad_bias, None, None, None


def _gather_sequence(x: torch.Tensor, tp=None, group=None) -> torch.Tensor:
    """[s / TP, b, c] -> [s, b, c] over the tensor-parallel group (layers.py:392-399: _all_gather_base along dim 0)."""
    if tp is None:
        tp, group = _tp()
    if tp == 1:
        return x
    out = torch.empty((x.shape[0] * tp,) + tuple(x.shape[1:]), dtype=x.dtype, device=x.device)
    dist.all_gather_into_tensor(out, x.contiguous(), group=group)
    return out


def _reduce_dgrad(grad_total: torch.Tensor, like: torch.Tensor, tp: int, group, allreduce_dgrad: bool, sequence_parallel: bool):
    """The communication half of a column-parallel linear's input gradient (layers.py:475-494)."""
    if allreduce_dgrad and tp > 1:
        dist.all_reduce(grad_total, group=group)
    if sequence_parallel:
        sub = torch.empty(like.shape, dtype=like.dtype, device=like.device)
        dist.reduce_scatter_tensor(sub, grad_total.contiguous(), group=group)
        return sub
    return grad_total


class NormLinearFn(torch.autograd.Function):
    """RMSNorm -> column-parallel linear as ONE autograd node (TELayerNormColumnParallelLinear's fusion, gpt_layer_specs.py:41): what is
    kept for the backward is the layer's own input x (the residual stream already holds it) instead of a second, normed [s, b, c]
    tensor per linear; the backward re-derives the normed rows with one vita_rmsnorm_fwd (HBM-bound, ~1 % of the layer).  Same kernels,
    same order, same bits as RMSNormFn followed by LinearFn."""

    @staticmethod
    def forward(ctx, x, ln_weight, weight, bias, eps, allreduce_dgrad, sequence_parallel):
        ctx.save_for_backward(x, ln_weight, weight)
        ctx.use_bias, ctx.eps = bias is not None, eps
        ctx.allreduce_dgrad, ctx.sequence_parallel = allreduce_dgrad, sequence_parallel
        ctx.tp, ctx.group = _tp()
        xn = ops.rmsnorm(x, ln_weight, eps)
        total = _gather_sequence(xn, ctx.tp, ctx.group) if sequence_parallel else xn
        s, b, c = total.shape
        out = ops.gemm(total.reshape(s * b, c), weight, ops.EPI_BIAS if bias is not None else ops.EPI_NONE, bias)
        return out.view(s, b, out.shape[-1])

    @staticmethod
    def backward(ctx, grad_output):
        x, ln_weight, weight = ctx.saved_tensors
        tp, group = ctx.tp, ctx.group
        xc = x.contiguous()
        xn = ops.rmsnorm(xc, ln_weight, ctx.eps)
        total = _gather_sequence(xn, tp, group) if ctx.sequence_parallel else xn
        s, b, c = total.shape
        go = grad_output.reshape(-1, grad_output.shape[-1]).contiguous()
        grad_total = dgrad(go, weight).view(s, b, c)
        d_xn = _reduce_dgrad(grad_total, x, tp, group, ctx.allreduce_dgrad, ctx.sequence_parallel)
        grad_weight, grad_bias = weight_bias_grads(go, total.reshape(s * b, c), ctx.needs_input_grad[2], ctx.use_bias,
                                                   weight.shape[0], weight.dtype)
        del total, xn
        d_lnw = torch.zeros(ln_weight.numel(), dtype=torch.float32, device=x.device)
        dx = ops.rmsnorm_bwd(d_xn.contiguous(), xc, ln_weight, ctx.eps, d_lnw)
        return dx.view_as(x), d_lnw.to(ln_weight.dtype), grad_weight, grad_bias, None, None, None


class GatedMLPFn(torch.autograd.Function):
    """The decoder's whole MLP — [RMSNorm ->] fc1 -> silu(gate) * up -> fc2 [-> TP reduction] — as ONE autograd node (Megatron MLP.forward,
    megatron/core/transformer/mlp.py, under stage 3's `--swiglu --disable-bias-linear`).  Kept for the backward: the input x (already
    held by the residual stream) and the fc1 product y [s, b, 2 ffn / TP]; the normed rows and the gated activation [s, b, ffn / TP] are
    re-derived (vita_rmsnorm_fwd, vita_swiglu_fwd: two HBM-bound passes, ~2 % of the layer's backward) — what training.TrainStep keeps
    for a layer outside the recompute block.  ln_weight None = the local spec (the norm is a separate module)."""

    @staticmethod
    def forward(ctx, x, ln_weight, w1, w2, eps, sequence_parallel):
        ctx.eps, ctx.sequence_parallel = eps, sequence_parallel
        ctx.has_norm = ln_weight is not None
        ctx.tp, ctx.group = _tp()
        xn = ops.rmsnorm(x, ln_weight, eps) if ctx.has_norm else x
        total = _gather_sequence(xn, ctx.tp, ctx.group) if sequence_parallel else xn
        s, b, c = total.shape
        y = ops.gemm(total.reshape(s * b, c), w1)
        act = ops.swiglu(y)
        out = ops.gemm(act, w2).view(s, b, w2.shape[0])
        del act
        if sequence_parallel:                                                  # RowParallelLinear :1095 / :1097
            sub = torch.empty((s // ctx.tp, b, w2.shape[0]), dtype=out.dtype, device=out.device)
            dist.reduce_scatter_tensor(sub, out, group=ctx.group)
            out = sub
        elif ctx.tp > 1:
            dist.all_reduce(out, group=ctx.group)
        ctx.save_for_backward(x, ln_weight, w1, w2, y)
        return out

    @staticmethod
    def backward(ctx, grad_output):
        x, ln_weight, w1, w2, y = ctx.saved_tensors
        tp, group, sp = ctx.tp, ctx.group, ctx.sequence_parallel
        go = _gather_sequence(grad_output.contiguous(), tp, group) if sp else grad_output          # ReduceScatterToSP / ReduceFromTP bwd
        go = go.reshape(-1, go.shape[-1]).contiguous()
        act = ops.swiglu(y)
        grad_w2, _ = weight_bias_grads(go, act, ctx.needs_input_grad[3], False, w2.shape[0], w2.dtype)
        del act
        d_act = dgrad(go, w2)
        dy = ops.swiglu_bwd(y, d_act)
        del d_act
        xc = x.contiguous()
        xn = ops.rmsnorm(xc, ln_weight, ctx.eps) if ctx.has_norm else xc
        total = _gather_sequence(xn, tp, group) if sp else xn
        s, b, c = total.shape
        grad_w1, _ = weight_bias_grads(dy, total.reshape(s * b, c), ctx.needs_input_grad[2], False, w1.shape[0], w1.dtype)
        del total, xn
        grad_total = dgrad(dy, w1).view(s, b, c)
        del dy
        d_xn = _reduce_dgrad(grad_total, x, tp, group, not sp, sp)
        if not ctx.has_norm:
            return d_xn.view_as(x), None, grad_w1, grad_w2, None, None
        d_lnw = torch.zeros(ln_weight.numel(), dtype=torch.float32, device=x.device)
        dx = ops.rmsnorm_bwd(d_xn.contiguous(), xc, ln_weight, ctx.eps, d_lnw)
        return dx.view_as(x), d_lnw.to(ln_weight.dtype), grad_w1, grad_w2, None, None


class ReduceFromTP(torch.autograd.Function):
    """reduce_from_tensor_model_parallel_region: all-reduce forward, identity backward (RowParallelLinear.forward :1097)."""

    @staticmethod
    def forward(ctx, x):
        tp, group = _tp()
        if tp > 1:
            x = x.contiguous()
            dist.all_reduce(x, group=group)
        return x

    @staticmethod
    def backward(ctx, g):
        return g


class ReduceScatterToSP(torch.autograd.Function):
    """reduce_scatter_to_sequence_parallel_region: reduce-scatter along dim 0 forward, all-gather backward (:1095)."""

    @staticmethod
    def forward(ctx, x):
        tp, group = ctx.tp, ctx.group = _tp()
        if tp == 1:
            return x
        out = torch.empty((x.shape[0] // tp,) + tuple(x.shape[1:]), dtype=x.dtype, device=x.device)
        dist.reduce_scatter_tensor(out, x.contiguous(), group=group)
        return out

    @staticmethod
    def backward(ctx, g):
        return _gather_sequence(g, ctx.tp, ctx.group)


class ScatterToSP(torch.autograd.Function):
    """scatter_to_sequence_parallel_region (language_model_embedding.py:157-160): this rank's s / TP rows forward, ALL-GATHER along the
    sequence backward — every tensor-parallel rank needs dL/d(embeddings) of the whole sequence: its vocabulary rows may be hit
    by tokens of any shard, and the replicated projector must see the same feature gradient on every rank."""

    @staticmethod
    def forward(ctx, x):
        tp, group = ctx.tp, ctx.group = _tp()
        if tp == 1:
            return x
        n = x.shape[0] // tp
        r = mpu.get_tensor_model_parallel_rank()
        return x[r * n:(r + 1) * n].contiguous()

    @staticmethod
    def backward(ctx, g):
        return _gather_sequence(g, ctx.tp, ctx.group)


class CopyToTP(torch.autograd.Function):
    """copy_to_tensor_model_parallel_region: identity forward, all-reduce backward (ColumnParallelLinear.forward :872)."""

    @staticmethod
    def forward(ctx, x):
        ctx.tp, ctx.group = _tp()
        return x

    @staticmethod
    def backward(ctx, g):
        tp, group = ctx.tp, ctx.group
        if tp > 1:
            g = g.contiguous()
            dist.all_reduce(g, group=group)
        return g


class GatherFromTP(torch.autograd.Function):
    """gather_from_tensor_model_parallel_region: all-gather along the last dim forward, split backward (:896-899)."""

    @staticmethod
    def forward(ctx, x):
        tp, group = _tp()
        ctx.tp, ctx.rank = tp, mpu.get_tensor_model_parallel_rank()
        if tp == 1:
            return x
        flat = torch.empty((tp,) + tuple(x.shape), dtype=x.dtype, device=x.device)
        dist.all_gather_into_tensor(flat.view(-1), x.contiguous().view(-1), group=group)
        return flat.movedim(0, -2).reshape(*x.shape[:-1], tp * x.shape[-1])

    @staticmethod
    def backward(ctx, g):
        tp, r = ctx.tp, ctx.rank
        if tp == 1:
            return g
        n = g.shape[-1] // tp
        return g[..., r * n:(r + 1) * n].contiguous()


# ------------------------------------------------------------------------------------------------
# norms
# ------------------------------------------------------------------------------------------------
class RMSNormFn(torch.autograd.Function):
    """RMSNorm.forward (M/core/transformer/custom_layers/transformer_engine.py:74-79) + vita_rmsnorm_bwd."""

    @staticmethod
    def forward(ctx, x, weight, eps):
        ctx.save_for_backward(x, weight)
        ctx.eps = eps
        return ops.rmsnorm(x, weight, eps)

    @staticmethod
    def backward(ctx, dy):
        x, weight = ctx.saved_tensors
        dw = torch.zeros(weight.numel(), dtype=torch.float32, device=x.device)
        xc = x.contiguous()
        dx = ops.rmsnorm_bwd(dy.contiguous(), xc, weight, ctx.eps, dw)
        return dx.view_as(x), dw.to(weight.dtype), None


class LayerNormFn(torch.autograd.Function):
    """torch.nn.LayerNorm / TENorm over the last dim (the InternViT block norms, M/core/models/vision/intern_vit_model.py:46,72):
    vita_layernorm_fwd + vita_layernorm_bwd (dx and both parameter gradients in one pass over the rows)."""

    @staticmethod
    def forward(ctx, x, weight, bias, eps):
        ctx.save_for_backward(x, weight)
        ctx.eps, ctx.has_bias = eps, bias is not None
        return ops.layernorm(x, weight, bias, eps)

    @staticmethod
    def backward(ctx, dy):
        x, weight = ctx.saved_tensors
        dg = torch.zeros(weight.numel(), dtype=torch.float32, device=x.device)
        db = torch.zeros_like(dg)
        dx = ops.layernorm_bwd(dy.contiguous(), x.contiguous(), weight, ctx.eps, dg, db)
        return dx.view_as(x), dg.to(weight.dtype), (db.to(weight.dtype) if ctx.has_bias else None), None


class GeluFn(torch.autograd.Function):
    """bf16(gelu(x)): erf form (`args.activation_func = torch.nn.functional.gelu`, M/pretrain_long_vita.py:207) or, `tanh=True`, the
    tanh approximation of SigLIP (`partial(F.gelu, approximate="tanh")`, :290): vita_gelu_fwd / vita_gelu_bwd / vita_gelu_tanh_bwd."""

    @staticmethod
    def forward(ctx, x, tanh=False):
        ctx.save_for_backward(x)
        ctx.tanh = bool(tanh)
        return ops.gelu(x.contiguous(), tanh=ctx.tanh)

    @staticmethod
    def backward(ctx, g):
        (x,) = ctx.saved_tensors
        return ops.gelu_bwd(x.contiguous(), g.contiguous(), tanh=ctx.tanh).view_as(x), None


class BiasScaleResidualFn(torch.autograd.Function):
    """`hidden = residual + (out + bias) * ls` of InternViTTransformerLayer.forward (intern_vit_model.py:60-66 / :79-82), one kernel
    with the module-by-module rounding chain; bias / scale may be None (SigLIP: no LayerScale)."""

    @staticmethod
    def forward(ctx, x, bias, scale, residual):
        ctx.save_for_backward(x, bias, scale)
        return ops.bias_scale_residual(x.contiguous(), bias, scale, residual.contiguous())

    @staticmethod
    def backward(ctx, g):
        x, bias, scale = ctx.saved_tensors
        g = g.contiguous()
        cols = x.shape[-1]
        f32 = lambda: torch.zeros(cols, dtype=torch.float32, device=g.device)  # noqa: E731
        d_bias = f32() if bias is not None and ctx.needs_input_grad[1] else None
        d_scale = f32() if scale is not None and ctx.needs_input_grad[2] else None
        dx = ops.bias_scale_residual_bwd(g, x.contiguous(), bias, scale, d_bias, d_scale)
        return (dx.view_as(x), None if d_bias is None else d_bias.to(bias.dtype), None if d_scale is None else d_scale.to(scale.dtype), g)


# ------------------------------------------------------------------------------------------------
# SwiGLU (bias-free gated linear unit of the decoder MLP: Megatron's MLP.forward glu closure)
# ------------------------------------------------------------------------------------------------
class SwiGLUFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, y):
        ctx.save_for_backward(y)
        two_f = y.shape[-1]
        return ops.swiglu(y.reshape(-1, two_f).contiguous()).view(*y.shape[:-1], two_f // 2)

    @staticmethod
    def backward(ctx, da):
        (y,) = ctx.saved_tensors
        two_f = y.shape[-1]
        return ops.swiglu_bwd(y.reshape(-1, two_f).contiguous(), da.reshape(-1, two_f // 2).contiguous()).view_as(y)


# ------------------------------------------------------------------------------------------------
# RoPE (apply_rotary_pos_emb_bshd, rotary_pos_embedding.py:181-204); the backward is the rotation by -theta
# ------------------------------------------------------------------------------------------------
class RopeFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, t, cos, sin):
        ctx.save_for_backward(cos, sin)
        out = t.contiguous().clone()
        s, b, h, d = out.shape
        ops.rope_apply_(out.view(s * b, h, d), cos, sin)
        return out

    @staticmethod
    def backward(ctx, g):
        cos, sin = ctx.saved_tensors
        g = g.contiguous().clone()
        s, b, h, d = g.shape
        ops.rope_apply_(g.view(s * b, h, d), cos, sin, sign=-1)
        return g, None, None


# ------------------------------------------------------------------------------------------------
# core attention (vita_flash_attn_fwd / vita_flash_attn_bwd)
# ------------------------------------------------------------------------------------------------
class FlashAttnFn(torch.autograd.Function):
    """CP = 1.  q [1, S, Hq, D], k / v [1, S, Hkv, D] -> [1, S, Hq, D]; seg_start / seg_end int32 [S] | None: packed samples
    (position ids with resets; _flash_attention_forward(position_ids=...), M/core/transformer/dot_product_attention.py:374-390)."""

    @staticmethod
    def forward(ctx, q, k, v, softmax_scale, causal=True, seg_start=None, seg_end=None, kept=None):
        """kept = (o, lse) of an earlier run on the same inputs (recompute_cache: the replay of a checkpointed layer): no kernel launch."""
        if not causal:
            raise ValueError("non-causal attention goes through FlashAttnNonCausalFn")
        if kept is not None:
            o, lse = kept
        else:
            o, lse = ops.flash_attn(q, k, v, causal=True, softmax_scale=softmax_scale, return_lse=True, seg_start=seg_start)
        ctx.save_for_backward(q, k, v, o, lse, seg_start, seg_end)
        ctx.softmax_scale = softmax_scale
        return o

    @staticmethod
    def backward(ctx, d_o):
        q, k, v, o, lse, seg_start, seg_end = ctx.saved_tensors
        dq, dk, dv = ops.flash_attn_bwd(q, k, v, o, d_o.contiguous(), lse, softmax_scale=ctx.softmax_scale,
                                        seg_start=seg_start, seg_end=seg_end)
        return dq, dk, dv, None, None, None, None, None


def non_causal_backward_plan(S: int, D: int, mode: str = "0"):
    """(padded sequence length, head size of the padded copies, head size of the dQ pass) of FlashAttnNonCausalFn.backward.
    d = 64: both general kernels at 64.  d = 96 (SigLIP, r05): measured at 64 frames x 16 heads x 1024 tokens — dQ (general kernel) 1.43 ms
    at 128 -> 1.20 ms at 96; dK + dV 1.30 ms at 128 (the pair kernel of attn_bwd_kvp.hip takes whole 256-row sequences at d = 128) against
    2.08 ms through the general kernel at 96.  So: copies padded to 128 for the pair kernel whenever it is eligible, the dQ pass reading the
    SAME buffers as d = 96 views (strides are free; columns 96 .. 127 are zero and are not read); otherwise both general kernels at 96.
    mode (VITA_VIT_BWD_PAD128, developer A / B): "1" = the r03 path, everything zero-padded to 128; "96" = never the pair kernel."""
    sp = -(-S // 128) * 128
    pair = D == 96 and sp % 256 == 0 and mode != "96"
    dp = 128 if (mode == "1" or D not in (64, 96) or pair) else D
    dq_d = D if (D in (64, 96) and mode != "1") else dp
    return sp, dp, dq_d


class FlashAttnNonCausalFn(torch.autograd.Function):
    """The ViT's core attention (flash_attn_func(causal=False), M/core/transformer/dot_product_attention.py:312-329) under autograd.
    q / k / v [B, S, H, D] views (B = frames, S = 1025, D = 64 for InternViT; SigLIP: 729 rows, D = 72 zero-padded to 96 by the caller).  Forward = the non-causal d = 64 kernel.  Backward =
    the general backward kernels at the NATIVE head size (r04: attn_bwd.hip's kernels are templates on d = 64 | 128, r05: | 96; through r03 the
    d = 64 tensors were zero-padded to 128: twice the flops and bytes), un-masked through their chunk tables (query chunk id 1 > key
    chunk id 0: every key is visible, there is no diagonal) on copies [S_pad, B * H, D] whose sequence is padded to a multiple of 128:
    frames x heads become the kernel's head index (Megatron's [s, b, np, hn] layout), padded KEYS have K = V = 0 (they add nothing to
    dQ, and their own dK / dV rows are dropped), padded QUERY rows carry lse = +inf (P = 0: they add nothing to dK / dV)."""

    @staticmethod
    def forward(ctx, q, k, v, softmax_scale):
        o, lse = ops.flash_attn(q, k, v, causal=False, softmax_scale=softmax_scale, return_lse=True)
        ctx.save_for_backward(q, k, v, o, lse)
        ctx.softmax_scale = softmax_scale
        return o

    @staticmethod
    def backward(ctx, d_o):
        q, k, v, o, lse = ctx.saved_tensors
        B, S, H, D = q.shape
        if k.shape[2] != H:
            raise NotImplementedError("the ViT attention backward is built for multi-head attention (ng == np)")
        import os
        sp, dp, dq_d = non_causal_backward_plan(S, D, os.environ.get("VITA_VIT_BWD_PAD128", "0"))
        nh = B * H

        def pad(t):                                   # [B, S, H, D] -> [1, S_pad, B * H, dp], zero-filled
            buf = torch.zeros(sp, B, H, dp, dtype=t.dtype, device=t.device)
            buf[:S, :, :, :D].copy_(t.transpose(0, 1))
            return buf.view(1, sp, nh, dp)

        qp, kp, vp, op_, dop = pad(q), pad(k), pad(v), pad(o), pad(d_o)
        lse_p = torch.full((1, nh, sp), float("inf"), dtype=torch.float32, device=q.device)
        lse_p[0, :, :S].copy_(lse.reshape(nh, S))
        scale = ctx.softmax_scale if ctx.softmax_scale is not None else 1.0 / (D ** 0.5)
        geo = dict(chunk_len=sp, q_chunk_gid=[1], kv_chunk_gid=[0], kv_chunk_row=[0], softmax_scale=scale)
        if dq_d == dp:
            dq, dk, dv = ops.flash_attn_bwd(qp, kp, vp, op_, dop, lse_p, **geo)
        else:
            _, dk, dv, delta = ops.flash_attn_bwd(qp, kp, vp, op_, dop, lse_p, parts=ops.ATTN_BWD_DKV, **geo)
            dq = torch.empty_like(qp)                 # columns dq_d .. dp - 1 are never written and never read (unpad takes :D)
            ops.flash_attn_bwd(qp[..., :dq_d], kp[..., :dq_d], vp[..., :dq_d], op_[..., :dq_d], dop[..., :dq_d], lse_p, parts=ops.ATTN_BWD_DQ,
                               delta=delta, dq5=dq[..., :dq_d], **geo)

        def unpad(t):
            return t.view(sp, B, H, dp)[:S, :, :, :D].transpose(0, 1)
        return unpad(dq), unpad(dk), unpad(dv), None


def zigzag_geometry(cp: int, rank: int, s_l: int) -> dict:
    """Chunk tables of the rank-ordered gathered K / V buffer [CP][2][S_l] rows (K rows of rank p at p * 2 * S_l, V at + S_l):
    chunk 2p + h of the buffer is global chunk (h ? 2CP - 1 - p : p) (M/training/utils.py:329-341)."""
    c = s_l // 2
    kv_gid, kv_row = [], []
    for p in range(cp):
        kv_gid += [p, 2 * cp - 1 - p]
        kv_row += [p * 2 * s_l, p * 2 * s_l + c]
    return dict(chunk_len=c, q_chunk_gid=mpu.zigzag_chunk_ids(cp, rank), kv_chunk_gid=kv_gid, kv_chunk_row=kv_row)


class FlashAttnCPFn(torch.autograd.Function):
    """Context-parallel causal attention as ONE autograd node — what TransformerEngine's AttnFuncWithCP is to the reference's TE
    layer spec (M/core/models/gpt/gpt_layer_specs.py:40; every long-context script trains with CP, stage3 .sh:125,145).

    q [1, S_l, Hq, D], k / v [1, S_l, Hkv, D]: the rank's two zig-zag chunks (rotated).  Forward =
    dot_product_attention.DotProductAttention.forward_cp (K / V all-gather per kv-head split over the CP group, own chunks first,
    chunk-table kernel) keeping O and the log-sum-exp over ALL keys.  Backward = K / V re-gathered (one message; keeping the
    gathered copy alive per layer would cost CP x the local K / V), vita_flash_attn_bwd writing dK / dV of every visible key in
    the gathered layout, ONE reduce-scatter returning each rank the sum for its own rows — the same two collectives
    training.TrainStep issues, instead of TE's CP - 1 P2P ring steps in each direction."""

    @staticmethod
    def run_forward(q, k, v, impl):
        """The forward proper -> (o [1, S_l, Hq, D], lse [1, Hq, S_l]); also what a checkpointed layer's first (no-grad) run calls when
        its result is kept for the replay (recompute_cache)."""
        _, s_l, hq, d = q.shape
        hkv = k.shape[2]
        if s_l % 2:
            raise ValueError("local sequence must hold two zig-zag chunks")
        n_split = ops.cp_kv_split(hkv, hq, s_l)                            # gather j + 1 runs under the attention of split j
        hg = hkv // n_split
        kv_local = torch.empty(n_split, 2, s_l, hg, d, dtype=q.dtype, device=q.device)
        kv_local[:, 0].copy_(k[0].reshape(s_l, n_split, hg, d).permute(1, 0, 2, 3))
        kv_local[:, 1].copy_(v[0].reshape(s_l, n_split, hg, d).permute(1, 0, 2, 3))
        lse = torch.empty(1, hq, s_l, dtype=torch.float32, device=q.device)
        o = torch.empty(1, s_l, hq, d, dtype=q.dtype, device=q.device)
        impl.forward_cp(q.reshape(1, s_l, hkv, hq // hkv, d), kv_local, out=o, lse=lse)
        return o, lse

    @staticmethod
    def forward(ctx, q, k, v, impl, kept=None):
        """kept = (o, lse) of an earlier run on the same inputs: neither the K / V all-gather nor the kernels run again."""
        cp, r, group = mpu.get_context_parallel_world_size(), mpu.get_context_parallel_rank(), mpu.get_context_parallel_group()
        o, lse = kept if kept is not None else FlashAttnCPFn.run_forward(q, k, v, impl)
        ctx.save_for_backward(q, k, v, o, lse)
        ctx.cp, ctx.rank, ctx.group, ctx.softmax_scale = cp, r, group, impl.softmax_scale   # the backward runs on autograd's thread
        return o

    @staticmethod
    def backward(ctx, d_o):
        """r04: the exchange runs under the kernels — the K / V re-gather is asynchronous per kv-head split (gather j + 1 in flight under
        split j's kernels), the dK + dV pass of a split runs BEFORE its dQ pass and its reduce-scatter is issued right behind it, so it
        runs under the dQ pass (3 of the backward's 8 GEMM units) and the later splits; every collective is waited for at the end.
        Through r03 this was: gather (sync) -> all kernels -> reduce-scatter (sync)."""
        q, k, v, o, lse = ctx.saved_tensors
        cp, r, group = ctx.cp, ctx.rank, ctx.group
        _, s_l, hq, d = q.shape
        hkv = k.shape[2]
        n_split = ops.cp_kv_split(hkv, hq, s_l)
        hg, qpg = hkv // n_split, hq // hkv
        d_o = d_o.contiguous()
        kv_local = torch.empty(n_split, 2, s_l, hg, d, dtype=q.dtype, device=q.device)        # the packed send buffers, one per split
        kv_local[:, 0].copy_(k[0].reshape(s_l, n_split, hg, d).permute(1, 0, 2, 3))
        kv_local[:, 1].copy_(v[0].reshape(s_l, n_split, hg, d).permute(1, 0, 2, 3))
        gathered = torch.empty(n_split, cp * kv_local[0].numel(), dtype=q.dtype, device=q.device)
        gathers = [dist.all_gather_into_tensor(gathered[j], kv_local[j].reshape(-1), group=group, async_op=True) for j in range(n_split)]
        d_rows = torch.empty_like(gathered)                                                  # dK rows of rank p at p * 2 * S_l, dV at + S_l
        dkv = torch.empty(n_split, kv_local[0].numel(), dtype=q.dtype, device=q.device)
        dq = torch.empty(q.shape, dtype=q.dtype, device=q.device)
        geo = zigzag_geometry(cp, r, s_l)
        q5, dq5 = q.view(1, s_l, hkv, qpg, d), dq.view(1, s_l, hkv, qpg, d)
        delta, reduces = None, []
        for j in range(n_split):
            if gathers[j] is not None:
                gathers[j].wait()
            rows, drows = gathered[j].view(cp * 2 * s_l, hg, d), d_rows[j].view(cp * 2 * s_l, hg, d)
            hs = slice(j * hg * qpg, (j + 1) * hg * qpg)
            args = (q5[:, :, j * hg:(j + 1) * hg], rows.unsqueeze(0), rows[s_l:].unsqueeze(0), o[:, :, hs], d_o[:, :, hs], lse[:, hs])
            if delta is None:            # row sums of dO * O for ALL heads, once: every split's passes read their head slice of it
                delta = torch.empty((hq, s_l), dtype=torch.float32, device=q.device)
                ops.attn_delta(o, d_o, delta)
            kw = dict(softmax_scale=ctx.softmax_scale, delta=delta[hs], **geo)
            ops.flash_attn_bwd(*args, dk=drows.unsqueeze(0), dv=drows[s_l:].unsqueeze(0), parts=ops.ATTN_BWD_DKV, **kw)
            reduces.append(dist.reduce_scatter_tensor(dkv[j], d_rows[j], group=group, async_op=True))
            ops.flash_attn_bwd(*args, dq5=dq5[:, :, j * hg:(j + 1) * hg], parts=ops.ATTN_BWD_DQ, **kw)
        for w in reduces:
            if w is not None:
                w.wait()
        dkv = dkv.view(n_split, 2, s_l, hg, d)
        dk = dkv[:, 0].permute(1, 0, 2, 3).reshape(1, s_l, hkv, d)
        dv = dkv[:, 1].permute(1, 0, 2, 3).reshape(1, s_l, hkv, d)
        return dq, dk, dv, None, None


# ------------------------------------------------------------------------------------------------
# embedding lookup + visual-token scatter (language_model_embedding.py:102-142)
# ------------------------------------------------------------------------------------------------
class EmbeddingScatterFn(torch.autograd.Function):
    """weight [V_local, h], ids [b * s] int64 (already made local; -1 = not on this rank), feats [N * L, h] | None,
    tgt / src int64 index vectors | None  ->  [b * s, h].  Backward: fp32 scatter-add into the weight gradient for the
    rows that kept their word embedding, a row gather for the features."""

    @staticmethod
    def forward(ctx, weight, ids, feats, tgt, src):
        n, h = ids.numel(), weight.shape[1]
        keep = ids >= 0
        if bool(keep.all()):
            we = ops.row_gather(weight, ids)
        else:                                   # vocab-parallel: rows of other ranks stay zero, the caller all-reduces
            we = torch.zeros(n, h, dtype=weight.dtype, device=weight.device)
            sel = ops.mask_to_index(keep)
            if sel.numel():
                ops.row_scatter_(we, sel, ops.row_gather(weight, ids[sel].contiguous()))
        if feats is not None and tgt is not None:
            ops.row_scatter_(we, tgt, feats, src)
        ctx.save_for_backward(ids, tgt, src)
        ctx.wshape, ctx.wdtype = tuple(weight.shape), weight.dtype
        ctx.fshape = None if feats is None else tuple(feats.shape)
        return we

    @staticmethod
    def backward(ctx, g):
        ids, tgt, src = ctx.saved_tensors
        g = g.contiguous()
        d_weight = d_feats = None
        if ctx.needs_input_grad[0]:
            tok = ids.clone()
            if tgt is not None:
                tok[tgt] = -1                                                      # overwritten rows: no embedding gradient
            acc = torch.zeros(ctx.wshape, dtype=torch.float32, device=g.device)
            ops.row_scatter_add_f32_(acc, tok, g)
            d_weight = acc.to(ctx.wdtype)
        if ctx.fshape is not None and ctx.needs_input_grad[2] and tgt is not None:
            d_feats = torch.zeros(ctx.fshape, dtype=g.dtype, device=g.device)
            rows = ops.row_gather(g, tgt)
            ops.row_scatter_(d_feats, src if src is not None else torch.arange(tgt.numel(), device=g.device), rows)
        return d_weight, None, d_feats, None, None


from . import tracing as _tracing                                   # noqa: E402
_tracing.instrument_functions(globals())                            # roctx ranges per Function under VITA_DEBUG (no-op otherwise)
