"""Training step (forward + backward) of the Long-VITA path — mirror of `forward_step` / `loss_func`
(M/pretrain_long_vita.py:778-869) with `GPTVLModel.forward(..., labels=, logit_mask=)`
(M/core/models/multimodal/gpt_vl_model.py:371-416) and the reference's activation-recompute policy
(`--recompute-granularity full --recompute-method block`, stage3 .sh:152-154).

The reference gets its backward from torch autograd over Megatron modules.  Here the backward is an
explicit reverse sweep over the same expressions, every arithmetic step a libvita_hip.so kernel:

  forward   the inference fast path (fused epilogues), keeping only each layer's input `h_l`;
  backward  per layer, recompute the layer from `h_l` with its intermediates, then
            dgrad = gemm(dY, W^T) (vita_transpose_bf16 of the weight feeds the NT GEMM), wgrad = gemm_tn(dY, X) (r03:
            vita_gemm_bf16_tn takes both operands contraction-major, no transposed copies of the activations),
            vita_swiglu_bwd, vita_rmsnorm_bwd (+ residual), vita_flash_attn_bwd, vita_rope_qkv_bwd;
  context parallelism: K/V all-gather in forward (and recompute), ONE reduce-scatter of the
            gathered-layout dK/dV per layer in backward (what TE's ring does in CP-1 P2P steps);
  loss      vocabulary cross-entropy on the logit-masked rows, instruction shift
            (gpt_vl_model.py:389-391), mean over all masked tokens of all CP ranks.

Frozen ViT (`--vision-model-freeze`, stage3 .sh:203): no gradient flows into the encoder; the
projector (pre-LayerNorm + 2-layer MLP) and the whole decoder are trained.

Gradient convention: `grads` hold d(mean token loss)/d(param) contributions of THIS rank; summing
them over the CP ranks (`allreduce_grads`) gives the full gradient.  This equals the reference's
(loss x CP) / tokens followed by Megatron's gradient average over the DPxCP group.
"""
from __future__ import annotations

from typing import Optional

import torch
import torch.distributed as dist

from . import ops, parallel_state as mpu, tracing, training_utils
from .gpt_vl_model import GPTVLModel


def _t(x: torch.Tensor) -> torch.Tensor:
    return ops.transpose(x)


def _dgrad(dy: torch.Tensor, w: torch.Tensor, out=None, epilogue=ops.EPI_NONE, residual=None) -> torch.Tensor:
    """grad_input = grad_output.matmul(weight)  (layers.py:444):  dy [M, N], w [N, K] -> [M, K]; whole-tile shapes read the weight
    contraction-major as it lies (vita_gemm_bf16_nn, r04), others transpose it first."""
    if epilogue == ops.EPI_NONE and residual is None and ops.gemm_nn_ok(dy, w) and (out is None or out.stride(1) == 1):
        return ops.gemm_nn(dy, w, out=out)
    return ops.gemm(dy, _t(w), epilogue, residual=residual, out=out)


def _wgrad(dy_t: torch.Tensor, x: torch.Tensor) -> torch.Tensor:
    """grad_weight = grad_output.t().matmul(total_input)  (layers.py:522-523): dy_t [N, M], x [M, K] -> [N, K]."""
    return ops.gemm(dy_t, _t(x))


def _wgrad_tn(dy: torch.Tensor, x: torch.Tensor) -> torch.Tensor:
    """grad_weight = grad_output.t().matmul(total_input) (layers.py:522-523) from the operands AS THE FORWARD LEFT THEM:
    dy [M, N], x [M, K] -> [N, K] through vita_gemm_bf16_tn (both operands contraction-major, no transposed copies);
    shapes the kernel does not tile (out dims not multiples of 256) take the two vita_transpose_bf16 passes + the NT GEMM."""
    if ops.gemm_tn_ok(dy, x):
        return ops.gemm_tn(dy, x)
    return ops.gemm(_t(dy), _t(x))


def _colsum(dy: torch.Tensor) -> torch.Tensor:
    """grad_bias = grad_output.sum(dim=0) (layers.py:524): dy [M, N] -> [N] in one pass (vita_colsum_bf16; r03: a GEMM against ones)."""
    return ops.colsum(dy).to(dy.dtype)


def _tp_sum(t: torch.Tensor) -> torch.Tensor:
    """Input gradient of a column-parallel linear: summed over the tensor-parallel group (layers.py:467-469)."""
    if mpu.get_tensor_model_parallel_world_size() > 1:
        dist.all_reduce(t, group=mpu.get_tensor_model_parallel_group())
    return t


def _pad_rows(x: torch.Tensor, mult: int = 64) -> torch.Tensor:
    m = x.shape[0]
    if m % mult == 0:
        return x
    out = torch.zeros((m + mult - m % mult,) + tuple(x.shape[1:]), dtype=x.dtype, device=x.device)
    out[:m] = x
    return out


class TrainStep:
    """forward_backward(tokens, labels, loss_mask, external_inputs) -> (loss, grads)."""

    def __init__(self, model: GPTVLModel, is_instruction_dataset: bool = True, recompute_num_layers: Optional[int] = None,
                 keep_attention: bool = False):
        """recompute_num_layers: Megatron's `--recompute-granularity full --recompute-method block --recompute-num-layers N`
        (M/training/arguments.py; stage 3 passes 20, stage 4 all layers): the FIRST N decoder layers keep only their input and are
        re-run in the backward, the others keep their activations.  None = every layer (what an 80 GB device needs at these
        sequence lengths); on 288 GB of HBM a 16K / 32K step keeps everything (N = 0: 1.5 GB per layer at 16K) and skips the
        second forward altogether.
        keep_attention (r04): the layers of the recompute block keep the ATTENTION half of their activations — the rotated qkv, the
        context, its log-sum-exp and the post-attention residual stream, 0.57 GB per 16K layer (27 GB for 48) — and re-derive only the
        two norms, the fc1 product and the SwiGLU in the backward: the second forward shrinks from 7.1 to 3.7 ms per 16K layer.  The
        same kernels produce the same values either way (bit-identical loss and gradients); it is what 288 GB of HBM are for, and it is
        off by default because Megatron's flag means "keep the layer input only"."""
        self.m = model
        self.is_instruction = is_instruction_dataset
        self.recompute_num_layers = recompute_num_layers
        self.keep_attention = bool(keep_attention)

    # ------------------------------------------------------------------------------------------
    def _layer_recompute(self, h, lp, cos, sin, mlp: bool = True):
        """Forward of one decoder layer from its input, keeping what the backward needs (mlp = False: up to the post-attention
        residual stream h_mid)."""
        m, c = self.m, self.m.cfg
        s = h.shape[0]
        cp = mpu.get_context_parallel_world_size()
        x1 = ops.rmsnorm(h, lp["ln1"], c.eps)
        qkv = ops.gemm(x1, lp["qkv_w"], ops.EPI_BIAS, lp["qkv_b"])
        kv_local = torch.empty(2, s, c.kv_groups, c.head_dim, dtype=h.dtype, device=h.device) if cp > 1 else None
        ops.rope_qkv_(qkv, c.kv_groups, c.qpg, c.head_dim, cos, sin, kv_local)
        m5 = qkv.view(1, s, c.kv_groups, c.qpg + 2, c.head_dim)
        q5 = m5[:, :, :, : c.qpg]
        if cp > 1:
            k_all, v_all, geo = self._gather_kv(kv_local)
        else:
            k_all, v_all, geo = m5[:, :, :, c.qpg], m5[:, :, :, c.qpg + 1], {}
        seg = training_utils.get_packed_segments()
        if seg is not None and cp > 1:
            raise NotImplementedError("packed samples under context parallelism are not built")
        ctx, lse = ops.flash_attn(q5, k_all, v_all, causal=True, return_lse=True, seg_start=None if seg is None else seg[0],
                                  **geo)
        ctx2 = ctx.view(s, c.heads * c.head_dim)
        tp = mpu.get_tensor_model_parallel_world_size()
        if tp == 1:
            h_mid = ops.gemm(ctx2, lp["o_w"], ops.EPI_RESIDUAL, residual=h)
        else:
            h_mid = m._row_parallel(ctx2, lp["o_w"], h.clone(), torch.empty_like(h))
        if not mlp:
            return dict(x1=x1, qkv=qkv, q5=q5, k_all=k_all, v_all=v_all, geo=geo, ctx=ctx, lse=lse, h_mid=h_mid)
        x2 = ops.rmsnorm(h_mid, lp["ln2"], c.eps)
        y = ops.gemm(x2, lp["fc1_w"])                      # unfused: the backward needs gate / up
        act = ops.swiglu(y)
        return dict(x1=x1, qkv=qkv, q5=q5, k_all=k_all, v_all=v_all, geo=geo, ctx=ctx, lse=lse, h_mid=h_mid, x2=x2,
                    y=y, act=act)

    def _layer_forward_keep(self, h, lp, cos, sin, light: bool = False):
        """Forward of one decoder layer that KEEPS its activations for the backward (a layer outside the recompute block):
        returns (layer output, what _layer_backward needs).  The cheap HBM-bound intermediates (both RMSNorm outputs, the
        SwiGLU product) are dropped again and re-derived in the backward; under CP the gathered K / V are re-gathered."""
        m = self.m
        if light:       # keep_attention: the attention half stays, the MLP runs with its SwiGLU epilogue and keeps nothing
            a = self._layer_recompute(h, lp, cos, sin, mlp=False)
            x2 = ops.rmsnorm(a["h_mid"], lp["ln2"], m.cfg.eps)
            act = ops.gemm(x2, lp["fc1_w"], ops.EPI_SWIGLU)
            if mpu.get_tensor_model_parallel_world_size() == 1:
                out = ops.gemm(act, lp["fc2_w"], ops.EPI_RESIDUAL, residual=a["h_mid"])
            else:
                out = m._row_parallel(act, lp["fc2_w"], a["h_mid"].clone(), torch.empty_like(h))
            return out, dict(qkv=a["qkv"], ctx=a["ctx"], lse=a["lse"], h_mid=a["h_mid"])
        a = self._layer_recompute(h, lp, cos, sin)
        if mpu.get_tensor_model_parallel_world_size() == 1:
            out = ops.gemm(a["act"], lp["fc2_w"], ops.EPI_RESIDUAL, residual=a["h_mid"])
        else:
            out = m._row_parallel(a["act"], lp["fc2_w"], a["h_mid"].clone(), torch.empty_like(h))
        keep = dict(qkv=a["qkv"], ctx=a["ctx"], lse=a["lse"], h_mid=a["h_mid"], y=a["y"])
        return out, keep

    def _rebuild(self, h, lp, keep):
        """The dictionary _layer_recompute returns, from the kept activations of _layer_forward_keep."""
        c = self.m.cfg
        s = h.shape[0]
        cp = mpu.get_context_parallel_world_size()
        qkv = keep["qkv"]
        m5 = qkv.view(1, s, c.kv_groups, c.qpg + 2, c.head_dim)
        if cp > 1:
            kv_local = torch.stack([m5[0, :, :, c.qpg], m5[0, :, :, c.qpg + 1]]).contiguous()      # rotated K, V of this rank
            k_all, v_all, geo = self._gather_kv(kv_local)
        else:
            k_all, v_all, geo = m5[:, :, :, c.qpg], m5[:, :, :, c.qpg + 1], {}
        x1 = ops.rmsnorm(h, lp["ln1"], c.eps)
        x2 = ops.rmsnorm(keep["h_mid"], lp["ln2"], c.eps)
        y = keep["y"] if "y" in keep else ops.gemm(x2, lp["fc1_w"])            # keep_attention: the fc1 product is re-derived
        return dict(x1=x1, qkv=qkv, q5=m5[:, :, :, : c.qpg], k_all=k_all, v_all=v_all, geo=geo, ctx=keep["ctx"], lse=keep["lse"],
                    h_mid=keep["h_mid"], x2=x2, y=y, act=ops.swiglu(y))

    def _gather_kv(self, kv_local):
        c = self.m.cfg
        cp, r = mpu.get_context_parallel_world_size(), mpu.get_context_parallel_rank()
        s_l = kv_local.shape[1]
        ch = s_l // 2
        gathered = torch.empty(cp * kv_local.numel(), dtype=kv_local.dtype, device=kv_local.device)
        dist.all_gather_into_tensor(gathered, kv_local.view(-1), group=mpu.get_context_parallel_group())
        rows = gathered.view(cp * 2 * s_l, c.kv_groups, c.head_dim)
        kv_gid, kv_row = [], []
        for p in range(cp):
            kv_gid += [p, 2 * cp - 1 - p]
            kv_row += [p * 2 * s_l, p * 2 * s_l + ch]
        geo = dict(chunk_len=ch, q_chunk_gid=mpu.zigzag_chunk_ids(cp, r), kv_chunk_gid=kv_gid, kv_chunk_row=kv_row)
        return rows.unsqueeze(0), rows[s_l:].unsqueeze(0), geo

    def _layer_backward(self, dh, h, lp, cos, sin, g, keep=None):
        """dh = dL/d(layer output) [s, hidden]; returns dL/d(layer input); fills g (this layer's grads).
        keep = the activations _layer_forward_keep stored, or None for a layer of the recompute block."""
        c = self.m.cfg
        s = h.shape[0]
        cp = mpu.get_context_parallel_world_size()
        a = self._layer_recompute(h, lp, cos, sin) if keep is None else self._rebuild(h, lp, keep)
        f32 = lambda n: torch.zeros(n, dtype=torch.float32, device=h.device)  # noqa: E731
        # ---- MLP: out = h_mid + fc2(swiglu(fc1(norm2(h_mid)))) ------------------------------------
        g["fc2_w"] = _wgrad_tn(dh, a["act"])
        d_act = _dgrad(dh, lp["fc2_w"])
        dy = ops.swiglu_bwd(a["y"], d_act)
        del d_act
        g["fc1_w"] = _wgrad_tn(dy, a["x2"])
        dx2 = _tp_sum(_dgrad(dy, lp["fc1_w"]))
        del dy
        dln2 = f32(c.hidden)
        dh_mid = ops.rmsnorm_bwd(dx2, a["h_mid"], lp["ln2"], c.eps, dln2, residual=dh)
        g["ln2"] = dln2
        del dx2
        # ---- attention: h_mid = h + proj(attn(rope(qkv(norm1(h))))) -------------------------------
        g["o_w"] = _wgrad_tn(dh_mid, a["ctx"].view(s, -1))
        d_ctx = _dgrad(dh_mid, lp["o_w"]).view(1, s, c.heads, c.head_dim)
        d_mixed = torch.empty_like(a["qkv"])
        dm5 = d_mixed.view(1, s, c.kv_groups, c.qpg + 2, c.head_dim)
        if cp > 1:
            dk_all = torch.empty_like(a["k_all"][0].reshape(-1)).view(cp * 2 * s, c.kv_groups, c.head_dim)
            # dK rows of rank p start at p*2*s, dV at +s: the layout of the gathered K/V buffer
            ops.flash_attn_bwd(a["q5"], a["k_all"], a["v_all"], a["ctx"], d_ctx, a["lse"], dq5=dm5[:, :, :, : c.qpg],
                               dk=dk_all.unsqueeze(0), dv=dk_all[s:].unsqueeze(0), **a["geo"])
            dkv_local = torch.empty(2 * s * c.kv_groups * c.head_dim, dtype=h.dtype, device=h.device)
            dist.reduce_scatter_tensor(dkv_local, dk_all.view(-1), group=mpu.get_context_parallel_group())
            dkv_local = dkv_local.view(2, s, c.kv_groups, c.head_dim)
            dm5[0, :, :, c.qpg].copy_(dkv_local[0])
            dm5[0, :, :, c.qpg + 1].copy_(dkv_local[1])
        else:
            seg = training_utils.get_packed_segments()
            ops.flash_attn_bwd(a["q5"], a["k_all"], a["v_all"], a["ctx"], d_ctx, a["lse"], dq5=dm5[:, :, :, : c.qpg],
                               dk=dm5[:, :, :, c.qpg], dv=dm5[:, :, :, c.qpg + 1],
                               seg_start=None if seg is None else seg[0], seg_end=None if seg is None else seg[1])
        ops.rope_qkv_bwd_(d_mixed, c.kv_groups, c.qpg, c.head_dim, cos, sin)
        g["qkv_w"] = _wgrad_tn(d_mixed, a["x1"])
        g["qkv_b"] = _colsum(d_mixed)                       # grad_bias = grad_output.sum(dim=0) (layers.py:524)
        dx1 = _tp_sum(_dgrad(d_mixed, lp["qkv_w"]))
        dln1 = f32(c.hidden)
        dh_in = ops.rmsnorm_bwd(dx1, h, lp["ln1"], c.eps, dln1, residual=dh_mid)
        g["ln1"] = dln1
        return dh_in

    # ------------------------------------------------------------------------------------------
    @torch.no_grad()
    def forward_backward(self, tokens: torch.Tensor, labels: torch.Tensor, loss_mask: torch.Tensor,
                         external_inputs: Optional[dict] = None, position_ids: Optional[torch.Tensor] = None):
        try:
            return self._forward_backward(tokens, labels, loss_mask, external_inputs, position_ids)
        finally:
            training_utils.set_position_ids(None)         # the reference's global (utils.py:44-50) must not leak

    def _forward_backward(self, tokens: torch.Tensor, labels: torch.Tensor, loss_mask: torch.Tensor,
                          external_inputs: Optional[dict] = None, position_ids: Optional[torch.Tensor] = None):
        """tokens / labels / loss_mask [1, S] (global, on every rank); external_inputs
        {"images", "indices"} (global).  `--logit-mask` semantics: logit_mask = loss_mask.bool().
        position_ids [1, S] with resets (--reset-position-ids): stage-2 packed samples — RoPE restarts and attention is
        block-diagonal (M/training/utils.py:267-270; CP = 1)."""
        m, c = self.m, self.m.cfg
        cp = mpu.get_context_parallel_world_size()
        seq = tokens.shape[1]
        training_utils.set_position_ids(None if position_ids is None else position_ids.transpose(0, 1).contiguous())
        if position_ids is None:
            position_ids = torch.arange(seq, dtype=torch.long, device=tokens.device).unsqueeze(0)
        batch = {"tokens": tokens, "labels": labels, "loss_mask": loss_mask, "position_ids": position_ids}
        if external_inputs:
            batch["external_images"] = external_inputs["images"]
            batch["external_indices"] = external_inputs["indices"]
        batch = training_utils.get_batch_on_this_cp_rank(batch, seq_length=seq) if cp > 1 else batch
        tok, lab, lmask = batch["tokens"], batch["labels"], batch["loss_mask"]
        logit_mask = lmask.bool()
        s = tok.shape[1]

        # ---- forward -------------------------------------------------------------------------------
        tracing.push("train: vision + embedding")
        proj = None
        efd = None
        # A CP rank whose two zig-zag chunks hold text only gets no src / tgt indices from get_batch_on_this_cp_rank
        # (M/training/utils.py:295,310-311).  The reference then runs the ViT over ALL frames and takes the
        # `features.mean() * 0` branch (language_model_embedding.py:132-134): nothing is scattered and the projector's
        # gradient is exactly zero.  Here that rank skips the encoder and contributes zero projector gradients, so the
        # collectives of the step (K/V gathers, reduce-scatters, allreduce_grads) stay consistent across ranks.
        text_only_rank = bool(external_inputs) and cp > 1 and "external_src_indices" not in batch
        if external_inputs and not text_only_rank:
            vis = m.external_feature_model
            images = batch["external_images"]
            vit_out = torch.cat([vis.vit(ch) for ch in torch.split(images, vis.cfg.chunk_frames, dim=0)], 0)
            proj = self._projector_forward(vis, vit_out)
            efd = {"features": proj["feats"]}
            if cp > 1:
                efd["src_indices"], efd["tgt_indices"] = batch["external_src_indices"], batch["external_tgt_indices"]
            else:
                efd["indices"] = batch["external_indices"]
        h = m.embedding(tok, None, external_feature_dict=efd).view(s, c.hidden)
        cos, sin = m.rotary_pos_emb(s * cp)
        ws = m._workspace(s, h.device)
        saved, kept = [], []
        n_layers = len(m.p["layers"])
        n_rec = n_layers if self.recompute_num_layers is None else max(0, min(n_layers, int(self.recompute_num_layers)))
        tracing.pop()
        for li, lp in enumerate(m.p["layers"]):
            tracing.push(f"train: fwd layer {li}")
            if li < n_rec and self.keep_attention:           # recompute block on 288 GB: only the MLP half is re-derived
                saved.append(h)
                h, keep = self._layer_forward_keep(h, lp, cos, sin, light=True)
                kept.append(keep)
            elif li < n_rec:                                 # recompute block: keep the input only, fused fast path
                saved.append(h.clone())
                kept.append(None)
                m.decoder_layer(h, lp, cos, sin, ws)
            else:                                            # activations stay in HBM: no second forward for this layer
                saved.append(h)
                h, keep = self._layer_forward_keep(h, lp, cos, sin)
                kept.append(keep)
            tracing.pop()
        tracing.push("train: head + loss")
        idx = ops.mask_to_index(logit_mask.transpose(0, 1).reshape(-1))
        n_sel = idx.numel()
        tp, tp_rank = mpu.get_tensor_model_parallel_world_size(), mpu.get_tensor_model_parallel_rank()
        # instruction shift (gpt_vl_model.py:389-391): logits[:-1] vs labels[1:]
        shift = 1 if self.is_instruction else 0
        n_loss = max(n_sel - shift, 0)
        stats = torch.zeros(2, dtype=torch.float32, device=h.device)
        # A context-parallel rank whose two zig-zag chunks hold no answer token selects nothing (config 5: the 512 answer tokens at
        # the end of a 128K row all sit in the last chunk, i.e. on CP rank 0): its head sees an empty [0, 1, hidden] input, contributes
        # zero loss and zero head gradients, and still takes part in every CP collective below.  (All TP ranks of a CP rank hold the
        # same tokens, so the TP collectives inside this block are entered by all of them or none.)
        if n_sel > 0:
            rows = ops.row_gather(h, idx)
            hn = ops.rmsnorm(rows, m.p["final_ln"], c.eps)
            hn_p = _pad_rows(hn)
            logits_local = ops.gemm(hn_p, m.p["lm_head"])                # [n_sel (padded), V / TP]
            logits = m._gather_vocab_parallel(logits_local).contiguous() # [n_sel (padded), V]
            ops.logit_postprocess_(logits, c.output_multiplier_scale, c.output_logit_softcapping)     # gpt_vl_model.py:349-355
            # labels of the selected rows (masked_select, gpt_vl_model.py:380-382); 16-byte rows for the gather
            lab_sel = ops.row_gather(lab.reshape(-1, 1).repeat(1, 2).contiguous(), idx)[:, 0]
            dlogits = torch.zeros_like(logits)
        if n_loss > 0:
            total = torch.tensor([float(n_loss)], dtype=torch.float32, device=h.device)
            if cp > 1:
                dist.all_reduce(total, group=mpu.get_context_parallel_group())
            scale = (1.0 / total).expand(n_loss).contiguous()
            loss_rows, dl = ops.ce_loss(logits[:n_loss], lab_sel[shift:shift + n_loss].contiguous(), scale, want_grad=True)
            dlogits[:n_loss] = dl
            stats[0], stats[1] = loss_rows.sum(), float(n_loss)
        elif cp > 1:
            dist.all_reduce(torch.zeros(1, dtype=torch.float32, device=h.device), group=mpu.get_context_parallel_group())
        if cp > 1:
            dist.all_reduce(stats, group=mpu.get_context_parallel_group())     # loss_func :801-803
        loss = stats[0] / stats[1].clamp(min=1.0)
        tracing.pop()

        # ---- backward ------------------------------------------------------------------------------
        grads = {"layers": [dict() for _ in m.p["layers"]]}
        dh = torch.zeros(s, c.hidden, dtype=h.dtype, device=h.device)
        if n_sel > 0:
            ops.logit_postprocess_bwd_(logits, dlogits, c.output_multiplier_scale, c.output_logit_softcapping)
            if tp > 1:      # this rank's vocabulary slice of dlogits
                v_l = logits_local.shape[1]
                dlogits = dlogits[:, tp_rank * v_l: (tp_rank + 1) * v_l].contiguous()
            grads["lm_head"] = _wgrad_tn(dlogits, hn_p)                      # [V / TP, hidden]
            d_hn = _tp_sum(_dgrad(dlogits, m.p["lm_head"]))[:n_sel]
            dfl = torch.zeros(c.hidden, dtype=torch.float32, device=h.device)
            d_rows = ops.rmsnorm_bwd(d_hn.contiguous(), rows, m.p["final_ln"], c.eps, dfl)
            grads["final_ln"] = dfl
            ops.row_scatter_(dh, idx, d_rows)                                # zeros.masked_scatter (layers.py:451)
        else:
            grads["lm_head"] = torch.zeros_like(m.p["lm_head"])
            grads["final_ln"] = torch.zeros(c.hidden, dtype=torch.float32, device=h.device)
        for li in range(len(m.p["layers"]) - 1, -1, -1):
            tracing.push(f"train: bwd layer {li}" + ("" if kept[li] is not None else " (recompute)"))
            dh = self._layer_backward(dh, saved[li], m.p["layers"][li], cos, sin, grads["layers"][li], kept[li])
            saved[li] = None
            kept[li] = None
            tracing.pop()
        tracing.push("train: embedding + projector + vision backward")
        # ---- embedding / visual-token scatter backward ---------------------------------------------
        tok_idx = tok.reshape(-1).clone()
        d_embed = torch.zeros(m.p["embed"].shape, dtype=torch.float32, device=h.device)
        if efd is not None:
            feats = proj["feats"]
            L = feats.shape[1]
            if "indices" in efd:
                ib, is_ = efd["indices"].unbind(dim=0)
                tgt = (ib.reshape(-1) * s + is_.reshape(-1))
                src = torch.arange(tgt.numel(), device=h.device)
            else:
                tgt = efd["tgt_indices"][0] * s + efd["tgt_indices"][1]
                src = efd["src_indices"][0] * L + efd["src_indices"][1]
            tok_idx[tgt] = -1                                           # overwritten rows: no embedding grad
            d_feats = torch.zeros(feats.shape[0] * L, c.hidden, dtype=h.dtype, device=h.device)
            ops.row_scatter_(d_feats, src, ops.row_gather(dh, tgt))
            self._projector_backward(m.external_feature_model, proj, d_feats, grads)
        elif text_only_rank:
            vp = m.external_feature_model.p
            grads["projector"] = {"proj_fc2": torch.zeros_like(vp["proj_fc2"]), "proj_fc1": torch.zeros_like(vp["proj_fc1"]),
                                  "proj_ln_w": torch.zeros(vp["proj_ln_w"].numel(), dtype=torch.float32, device=h.device),
                                  "proj_ln_b": torch.zeros(vp["proj_ln_b"].numel(), dtype=torch.float32, device=h.device)}
        ops.row_scatter_add_f32_(d_embed, tok_idx, dh)
        grads["embed"] = d_embed
        tracing.pop()
        return loss, grads

    # ------------------------------------------------------------------------------------------
    def _projector_forward(self, vis, vit_out):
        p, cfg = vis.p, vis.cfg
        ones = torch.ones_like(p["proj_ln_w"])
        xhat = ops.pixel_shuffle_ln(vit_out, ones, None, cfg.grid, cfg.add_class_token, cfg.proj_ln_eps)   # normalised, w=1 b=0
        t = ops.pixel_shuffle_ln(vit_out, p["proj_ln_w"], p["proj_ln_b"], cfg.grid, cfg.add_class_token, cfg.proj_ln_eps)
        t2 = t.view(-1, t.shape[-1])
        f1 = ops.gemm(t2, p["proj_fc1"])                                # pre-GELU, unfused
        a = ops.gemm(t2, p["proj_fc1"], ops.EPI_BIAS_GELU)
        feats = ops.gemm(a, p["proj_fc2"]).view(vit_out.shape[0], -1, cfg.llm_hidden)
        return dict(xhat=xhat.view(-1, xhat.shape[-1]), t=t2, f1=f1, a=a, feats=feats)

    def _projector_backward(self, vis, proj, d_feats, grads):
        p, cfg = vis.p, vis.cfg
        df_t = _t(_pad_rows(d_feats))
        g = {}
        g["proj_fc2"] = ops.gemm(df_t, _t(_pad_rows(proj["a"])))
        d_a = _dgrad(d_feats, p["proj_fc2"])
        d_f1 = ops.gelu_bwd(proj["f1"], d_a)
        g["proj_fc1"] = ops.gemm(_t(_pad_rows(d_f1)), _t(_pad_rows(proj["t"])))
        d_t = _dgrad(d_f1, p["proj_fc1"])
        dgam = torch.zeros(p["proj_ln_w"].numel(), dtype=torch.float32, device=d_feats.device)
        dbet = torch.zeros_like(dgam)
        ops.layernorm_param_grad(d_t, proj["xhat"], dgam, dbet, cfg.proj_ln_eps, prenormalized=True)
        g["proj_ln_w"], g["proj_ln_b"] = dgam, dbet
        grads["projector"] = g


def allreduce_grads(grads) -> None:
    """Sum the per-rank gradient contributions over the context-parallel group (fp32 on the wire)."""
    if mpu.get_context_parallel_world_size() == 1:
        return
    group = mpu.get_context_parallel_group()

    def walk(node):
        if isinstance(node, dict):
            for k, v in node.items():
                node[k] = walk(v)
            return node
        if isinstance(node, list):
            return [walk(v) for v in node]
        t = node.float()
        dist.all_reduce(t, group=group)
        return t.to(node.dtype)

    walk(grads)
