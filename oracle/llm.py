"""Qwen2.5-style decoder restated in Megatron weight layout.
TEST INFRASTRUCTURE ONLY."""
from __future__ import annotations

from dataclasses import dataclass

import torch
import torch.nn.functional as F

from . import glue
from .attention import core_attention, zigzag_cp_attention
from .vit import linear


@dataclass
class LLMConfig:
    """stage3 .sh:155-200 / H/models/long_vita_qwen2_intern/config_14B.json:31-56."""
    num_layers: int = 48
    hidden: int = 5120
    heads: int = 40
    kv_groups: int = 8
    head_dim: int = 128
    ffn: int = 13824
    vocab: int = 152064
    eps: float = 1e-6
    rope_theta: float = 1e6

    @property
    def qpg(self):
        return self.heads // self.kv_groups

    @property
    def qkv_out(self):
        return (self.heads + 2 * self.kv_groups) * self.head_dim


def init_llm_params(cfg: LLMConfig, seed: int = 1234, dtype=torch.bfloat16, std: float = 0.02):
    """Seeded synthetic weights, MEGATRON layout: linear_qkv rows per group [q x qpg, k, v]
    (R/tools/hf2mcore_long_vita.py:597-609), linear_fc1 = cat[gate, up] (:612)."""
    g = torch.Generator().manual_seed(seed)

    def rn(*shape, s=std):
        return (torch.randn(*shape, generator=g) * s).to(dtype)

    p = {"embed": rn(cfg.vocab, cfg.hidden), "layers": [], "final_ln": torch.ones(cfg.hidden, dtype=dtype),
         "lm_head": rn(cfg.vocab, cfg.hidden)}
    for _ in range(cfg.num_layers):
        p["layers"].append({
            "ln1": torch.ones(cfg.hidden, dtype=dtype),
            "qkv_w": rn(cfg.qkv_out, cfg.hidden), "qkv_b": rn(cfg.qkv_out),
            "o_w": rn(cfg.hidden, cfg.heads * cfg.head_dim),
            "ln2": torch.ones(cfg.hidden, dtype=dtype),
            "fc1_w": rn(2 * cfg.ffn, cfg.hidden),
            "fc2_w": rn(cfg.hidden, cfg.ffn),
        })
    return p


def split_qkv(mixed, cfg: LLMConfig):
    """Megatron SelfAttention.get_query_key_value_tensors (restated at
    L/core/models/vision/intern_vit_model.py:145-197): [s, b, ng*(qpg+2)*d] -> q [s,b,np,d], k, v [s,b,ng,d]."""
    s, b, _ = mixed.shape
    m = mixed.view(s, b, cfg.kv_groups, (cfg.qpg + 2) * cfg.head_dim)
    q, k, v = torch.split(m, [cfg.qpg * cfg.head_dim, cfg.head_dim, cfg.head_dim], dim=3)
    return q.reshape(s, b, cfg.heads, cfg.head_dim), k, v


def decoder_layer(h, lp, cfg: LLMConfig, freqs, attn_fn):
    """TE layer spec order (M/core/models/gpt/gpt_layer_specs.py:35-49): RMSNorm -> QKV(+bias) ->
    RoPE -> core attention -> proj -> residual; RMSNorm -> fc1 -> SwiGLU -> fc2 -> residual.
    h [s, b, hidden]; freqs [s, 1|b, 1, d] fp32; attn_fn(q, k, v) -> [s, b, hidden]."""
    res = h
    x = glue.rmsnorm(h, lp["ln1"], cfg.eps)
    q, k, v = split_qkv(linear(x, lp["qkv_w"], lp["qkv_b"]), cfg)
    q = glue.apply_rotary_pos_emb_bshd(q, freqs)
    k = glue.apply_rotary_pos_emb_bshd(k, freqs)
    ctx = attn_fn(q, k, v)
    h = res + linear(ctx, lp["o_w"])
    res = h
    x = glue.rmsnorm(h, lp["ln2"], cfg.eps)
    y = linear(x, lp["fc1_w"])
    gate, up = torch.chunk(y, 2, dim=-1)
    h = res + linear(F.silu(gate.float()).to(y.dtype) * up, lp["fc2_w"])
    return h, (k, v)


def prefill_logits(input_ids, p, cfg: LLMConfig, logit_positions, external_feature_dict=None, position_ids=None):
    """GPTVLModel.forward with labels=None, CP=1 (M/core/models/multimodal/gpt_vl_model.py:233-416):
    embedding(+scatter) -> layers -> final norm -> logits-masked head.  input_ids [1, S]."""
    S = input_ids.shape[1]
    we = p["embed"][input_ids]                                   # [b, s, h]
    h = glue.embedding_scatter(we, external_feature_dict)         # [s, b, h]
    pos = position_ids.transpose(0, 1) if position_ids is not None else None
    freqs = glue.rope_emb(S, glue.rope_inv_freq(cfg.head_dim, cfg.rope_theta), pos)
    for lp in p["layers"]:
        h, _ = decoder_layer(h, lp, cfg, freqs, lambda q, k, v: core_attention(q, k, v, causal=True))
    h = glue.rmsnorm(h, p["final_ln"], cfg.eps)
    mask = torch.zeros(1, S, dtype=torch.bool)
    mask[0, list(logit_positions)] = True
    logits = glue.masked_linear_fwd(h.float(), p["lm_head"].float(), None, mask)   # [n_sel, b, V] fp32
    return logits.transpose(0, 1).contiguous()


def prefill_logits_cp(input_ids, p, cfg: LLMConfig, cp_size: int, logit_positions_per_rank, features_per_rank=None):
    """The same prefill sharded zig-zag over cp_size ranks, all ranks simulated in one process:
    per layer every rank computes its local QKV, then attends over the K/V of all ranks.
    Returns per-rank logits [1, n_sel, V]."""
    S = input_ids.shape[1]
    inv = glue.rope_inv_freq(cfg.head_dim, cfg.rope_theta)
    hs, freqs = [], []
    for r in range(cp_size):
        ids = glue.zigzag_slice(input_ids, cp_size, r)
        we = p["embed"][ids]
        hs.append(glue.embedding_scatter(we, None if features_per_rank is None else features_per_rank[r]))
        freqs.append(glue.rope_emb(S, inv, None, cp_size, r))
    for lp in p["layers"]:
        qs, ks, vs, ress = [], [], [], []
        for r in range(cp_size):
            x = glue.rmsnorm(hs[r], lp["ln1"], cfg.eps)
            q, k, v = split_qkv(linear(x, lp["qkv_w"], lp["qkv_b"]), cfg)
            qs.append(glue.apply_rotary_pos_emb_bshd(q, freqs[r]))
            ks.append(glue.apply_rotary_pos_emb_bshd(k, freqs[r]))
            vs.append(v)
        for r in range(cp_size):
            ctx = zigzag_cp_attention(qs[r], ks, vs, cp_size, r, S)
            h = hs[r] + linear(ctx, lp["o_w"])
            x = glue.rmsnorm(h, lp["ln2"], cfg.eps)
            y = linear(x, lp["fc1_w"])
            gate, up = torch.chunk(y, 2, dim=-1)
            hs[r] = h + linear(F.silu(gate.float()).to(y.dtype) * up, lp["fc2_w"])
    outs = []
    for r in range(cp_size):
        h = glue.rmsnorm(hs[r], p["final_ln"], cfg.eps)
        mask = torch.zeros(1, h.shape[0], dtype=torch.bool)
        mask[0, list(logit_positions_per_rank[r])] = True
        outs.append(glue.masked_linear_fwd(h.float(), p["lm_head"].float(), None, mask).transpose(0, 1).contiguous())
    return outs


def to_hf_state_dict(p, cfg: LLMConfig):
    """Inverse of R/tools/hf2mcore_long_vita.py:597-613 — Megatron layout -> transformers Qwen2 names."""
    sd = {"model.embed_tokens.weight": p["embed"], "model.norm.weight": p["final_ln"], "lm_head.weight": p["lm_head"]}
    d, ng, qpg = cfg.head_dim, cfg.kv_groups, cfg.qpg
    for i, lp in enumerate(p["layers"]):
        w = lp["qkv_w"].view(ng, (qpg + 2) * d, cfg.hidden)
        b = lp["qkv_b"].view(ng, (qpg + 2) * d)
        pre = f"model.layers.{i}."
        sd[pre + "self_attn.q_proj.weight"] = w[:, : qpg * d].reshape(-1, cfg.hidden)
        sd[pre + "self_attn.k_proj.weight"] = w[:, qpg * d: (qpg + 1) * d].reshape(-1, cfg.hidden)
        sd[pre + "self_attn.v_proj.weight"] = w[:, (qpg + 1) * d:].reshape(-1, cfg.hidden)
        sd[pre + "self_attn.q_proj.bias"] = b[:, : qpg * d].reshape(-1)
        sd[pre + "self_attn.k_proj.bias"] = b[:, qpg * d: (qpg + 1) * d].reshape(-1)
        sd[pre + "self_attn.v_proj.bias"] = b[:, (qpg + 1) * d:].reshape(-1)
        sd[pre + "self_attn.o_proj.weight"] = lp["o_w"]
        gate, up = torch.chunk(lp["fc1_w"], 2, dim=0)
        sd[pre + "mlp.gate_proj.weight"] = gate
        sd[pre + "mlp.up_proj.weight"] = up
        sd[pre + "mlp.down_proj.weight"] = lp["fc2_w"]
        sd[pre + "input_layernorm.weight"] = lp["ln1"]
        sd[pre + "post_attention_layernorm.weight"] = lp["ln2"]
    return sd
