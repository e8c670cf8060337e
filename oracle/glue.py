"""Restatements of the first-party index / elementwise glue.
TEST INFRASTRUCTURE ONLY — see oracle/__init__.py."""
from __future__ import annotations

import torch


# --------------------------------------------------------------------------------------------
# a1 — zig-zag context-parallel slice + image/index remap
# --------------------------------------------------------------------------------------------
def index_of_a_in_b(a: torch.Tensor, b: torch.Tensor) -> torch.Tensor:
    """M/training/utils.py:347-350."""
    b_indices = torch.where(torch.isin(b, a))[0]
    b_values = b[b_indices]
    return b_indices[b_values.argsort()[a.argsort().argsort()]]


def zigzag_chunk_ids(cp_size: int, cp_rank: int):
    """Rank r owns chunks r and 2CP-1-r of the 2CP-chunk view (M/training/utils.py:329-341)."""
    return [cp_rank, 2 * cp_size - cp_rank - 1]


def calibration_index(seq_length: int, cp_size: int, cp_rank: int) -> torch.Tensor:
    """M/training/utils.py:281 — global positions owned by the rank, in local order."""
    return torch.arange(seq_length).view(2 * cp_size, seq_length // (2 * cp_size))[zigzag_chunk_ids(cp_size, cp_rank)].view(-1)


def zigzag_slice(val: torch.Tensor, cp_size: int, cp_rank: int, seq_dim: int = 1) -> torch.Tensor:
    """M/training/utils.py:329-341 (view as 2CP chunks, index_select, flatten)."""
    val = val.view(*val.shape[0:seq_dim], 2 * cp_size, val.shape[seq_dim] // (2 * cp_size), *val.shape[(seq_dim + 1):])
    index = torch.tensor(zigzag_chunk_ids(cp_size, cp_rank))
    val = val.index_select(seq_dim, index)
    return val.view(*val.shape[0:seq_dim], -1, *val.shape[(seq_dim + 2):])


def get_batch_on_this_cp_rank(batch: dict, seq_length: int, cp_size: int, cp_rank: int) -> dict:
    """M/training/utils.py:252-343 with args/mpu made explicit.  Keys: any [b, s, ...] tensors,
    plus optional external_images [N, ...] and external_indices [2, N, L]."""
    batch = dict(batch)
    if cp_size <= 1:
        return batch
    for key, val in list(batch.items()):
        if key == "external_images":
            if "external_indices" in batch:
                cal = calibration_index(seq_length, cp_size, cp_rank)
                indices_b, indices_s = batch["external_indices"].unbind(dim=0)
                mask = torch.isin(indices_s, cal)
                if mask.any():
                    selected_i = torch.any(mask, dim=1)
                    batch["external_images"] = val[selected_i, ...]
            continue
        if key == "external_indices":
            cal = calibration_index(seq_length, cp_size, cp_rank)
            indices_b, indices_s = val.unbind(dim=0)
            mask = torch.isin(indices_s, cal)
            if mask.any():
                selected_i = torch.any(mask, dim=1)
                num_images = int(selected_i.sum())
                src_b = torch.arange(num_images).unsqueeze(1).repeat(1, indices_b.shape[1])
                src_s = torch.arange(indices_s.shape[1]).unsqueeze(0).repeat(num_images, 1)
                src_b = src_b[mask[selected_i]]
                src_s = src_s[mask[selected_i]]
                tgt_b = indices_b[mask]
                tgt_s = index_of_a_in_b(indices_s[mask], cal)
                batch["external_src_indices"] = torch.stack([src_b, src_s])
                batch["external_tgt_indices"] = torch.stack([tgt_b, tgt_s])
            batch.pop(key)
            continue
        if key == "attention_mask":
            continue
        if val is not None:
            batch[key] = zigzag_slice(val, cp_size, cp_rank, seq_dim=1)
    return batch


# --------------------------------------------------------------------------------------------
# a8 — RoPE
# --------------------------------------------------------------------------------------------
def rope_inv_freq(dim: int, base: float) -> torch.Tensor:
    """rotary_pos_embedding.py:74-80."""
    return 1.0 / (base ** (torch.arange(0, dim, 2, dtype=torch.float32) / dim))


def rope_emb(max_seq_len: int, inv_freq: torch.Tensor, position_ids: torch.Tensor = None, cp_size: int = 1,
             cp_rank: int = 0) -> torch.Tensor:
    """RotaryEmbedding.forward, rotary_pos_embedding.py:84-122 -> emb [s, b|1, 1, dim] fp32.
    position_ids (if given) is [s, b] as stored by set_position_ids (M/training/utils.py:267-270)."""
    seq = torch.arange(max_seq_len, dtype=inv_freq.dtype)
    freqs = torch.outer(seq, inv_freq)
    emb = torch.cat((freqs, freqs), dim=-1)[:, None, None, :]
    if position_ids is not None:
        s, b = position_ids.shape
        emb = emb[position_ids.reshape(-1)].squeeze(1).reshape(s, b, 1, -1)
    if cp_size > 1:
        emb = zigzag_slice(emb, cp_size, cp_rank, seq_dim=0)      # get_pos_emb_on_this_cp_rank :36-47
    return emb


def rotate_half(x: torch.Tensor) -> torch.Tensor:
    """rotary_pos_embedding.py:169-171 (non-interleaved)."""
    x1, x2 = torch.chunk(x, 2, dim=-1)
    return torch.cat((-x2, x1), dim=-1)


def apply_rotary_pos_emb_bshd(t: torch.Tensor, freqs: torch.Tensor) -> torch.Tensor:
    """rotary_pos_embedding.py:181-204: cos/sin cast to t.dtype, arithmetic in t.dtype."""
    rot_dim = freqs.shape[-1]
    t, t_pass = t[..., :rot_dim], t[..., rot_dim:]
    cos_ = torch.cos(freqs).to(t.dtype)
    sin_ = torch.sin(freqs).to(t.dtype)
    t = (t * cos_) + (rotate_half(t) * sin_)
    return torch.cat((t, t_pass), dim=-1)


# --------------------------------------------------------------------------------------------
# a9 — RMSNorm
# --------------------------------------------------------------------------------------------
def rmsnorm(x: torch.Tensor, weight: torch.Tensor, eps: float = 1e-6) -> torch.Tensor:
    """M/core/transformer/custom_layers/transformer_engine.py:74-79."""
    xf = x.float()
    out = (xf * torch.rsqrt(xf.pow(2).mean(-1, keepdim=True) + eps)).type_as(x)
    return out * weight


# --------------------------------------------------------------------------------------------
# a7 — embedding + visual-token scatter
# --------------------------------------------------------------------------------------------
def embedding_scatter(word_embeddings: torch.Tensor, external_feature_dict: dict = None) -> torch.Tensor:
    """language_model_embedding.py:102-142: word_embeddings [b, s, h] -> [s, b, h]."""
    we = word_embeddings
    if external_feature_dict is not None:
        we = we.clone()
        features = external_feature_dict["features"]
        if "indices" in external_feature_dict:
            ib, is_ = external_feature_dict["indices"].unbind(dim=0)
            we[ib.view(-1), is_.view(-1)] = features.view(-1, features.shape[-1])
        elif "pre_len" in external_feature_dict:
            pre_len = external_feature_dict["pre_len"]
            we[:, pre_len:pre_len + features.shape[1]] = features
        elif "src_indices" in external_feature_dict and "tgt_indices" in external_feature_dict:
            sb, ss = external_feature_dict["src_indices"]
            tb, ts = external_feature_dict["tgt_indices"]
            we[tb, ts] = features[sb, ss]
        else:
            we = we + features.mean() * 0
    return we.transpose(0, 1).contiguous()


# --------------------------------------------------------------------------------------------
# a12 — logits-masked linear
# --------------------------------------------------------------------------------------------
def masked_linear_fwd(inp: torch.Tensor, weight: torch.Tensor, bias, logit_mask: torch.Tensor) -> torch.Tensor:
    """M/core/tensor_parallel/layers.py:402-412: inp [s, b, c], logit_mask [b, s] bool."""
    total_input = inp
    if logit_mask is not None:
        b, c = inp.size(1), inp.size(2)
        total_input = torch.masked_select(inp, logit_mask.transpose(0, 1).unsqueeze(2)).reshape(-1, b, c)
    out = torch.matmul(total_input, weight.t())
    if bias is not None:
        out = out + bias
    return out


def logit_postprocess(logits: torch.Tensor, output_multiplier_scale=None, output_logit_softcapping=None) -> torch.Tensor:
    """M/core/models/multimodal/gpt_vl_model.py:349-355, verbatim op order (every op rounds in logits.dtype)."""
    if output_multiplier_scale:
        logits = logits * output_multiplier_scale
    if output_logit_softcapping:
        logits = logits / output_logit_softcapping
        logits = torch.tanh(logits)
        logits = logits * output_logit_softcapping
    return logits


def masked_linear_bwd(grad_output: torch.Tensor, inp: torch.Tensor, weight: torch.Tensor, logit_mask: torch.Tensor):
    """layers.py:444-455,522-523: returns (grad_input [s,b,c], grad_weight [out,c])."""
    total_input = inp
    grad_input = grad_output.matmul(weight)
    if logit_mask is not None:
        s, b, c = logit_mask.size(1), grad_input.size(1), grad_input.size(2)
        grad_input = torch.zeros([s, b, c], dtype=grad_input.dtype).masked_scatter(
            logit_mask.transpose(0, 1).unsqueeze(2), grad_input)
        total_input = torch.masked_select(total_input, logit_mask.transpose(0, 1).unsqueeze(2)).reshape(-1, b, c)
    go = grad_output.reshape(-1, grad_output.shape[-1])
    ti = total_input.reshape(-1, total_input.shape[-1])
    return grad_input, go.t().matmul(ti)


# --------------------------------------------------------------------------------------------
# a13 — decode-time logit-mask rule and CP logits re-assembly
# --------------------------------------------------------------------------------------------
def loss_func(losses_per_rank, masks_per_rank, is_instruction: bool):
    """M/pretrain_long_vita.py:778-838 for all CP ranks at once (DP = 1).  Per rank: [sum(loss * mask), sum(mask)] with the
    mask shifted by one under --is-instruction-dataset (:793-796); the pair is all-reduced over the CP group (:801-803); the
    rank returns (loss_sum * CP, token count, reporting pair) (:833-838).  With --logit-mask the caller passes a ones mask one
    wider than the selected rows (forward_step :866-867), i.e. every selected row but the shifted-out one counts."""
    cp = len(losses_per_rank)
    parts = []
    for losses, mask in zip(losses_per_rank, masks_per_rank):
        m = (mask[..., 1:] if is_instruction else mask).reshape(-1).float()
        parts.append(torch.cat([torch.sum(losses.float().view(-1) * m).view(1), m.sum().view(1)]))
    tot = parts[0].clone()
    for p_ in parts[1:]:
        tot = tot + p_
    return tot[0] * cp, tot[1].to(torch.int), (tot[0], tot[1])


def cp_logit_mask_positions(context_length: int, local_len: int, cp_size: int, reference_compat: bool = True):
    """M/inference/text_generation/generation.py:141-165,195.  Returns (sorted local positions marked
    True in logit_mask, block index picked from the all-gathered [b, 2*CP, V] logits).
    reference_compat=False applies the fix for SURVEY.md §9 quirk 2 (token ctx-1 lives in block
    (ctx-1)//half; the reference picks ctx//half and wraps index -1 when ctx % half == 0)."""
    if cp_size == 1:
        return [context_length - 1], None
    half = local_len // 2
    if reference_compat:
        ccl = context_length % half
        block = context_length // half
        p0 = (ccl - 1) % local_len          # python negative index: -1 -> last local position
        p1 = half + ccl - 1
        return sorted([p0, p1]), block
    ccl = (context_length - 1) % half
    return [ccl, half + ccl], (context_length - 1) // half


def sync_output_order(cp_size: int):
    """generation.py:542-566: after all-gather, halves are ordered by sorted chunk id.
    Returns the permutation applied to the 2*CP gathered halves."""
    ids = []
    for r in range(cp_size):
        ids += [r, 2 * cp_size - r - 1]
    return torch.sort(torch.tensor(ids)).indices.tolist()


# --------------------------------------------------------------------------------------------
# a5 — pixel shuffle
# --------------------------------------------------------------------------------------------
def pixel_shuffle(x: torch.Tensor, scale_factor: float = 0.5) -> torch.Tensor:
    """M/pretrain_long_vita.py:572-582 ≡ H/.../resampler_projector.py:36-46."""
    n, w, h, c = x.size()
    x = x.view(n, w, int(h * scale_factor), int(c / scale_factor))
    x = x.permute(0, 2, 1, 3).contiguous()
    x = x.view(n, int(h * scale_factor), int(w * scale_factor), int(c / (scale_factor * scale_factor)))
    x = x.permute(0, 2, 1, 3).contiguous()
    return x
