"""GPTVLModel — mirror of M/core/models/multimodal/gpt_vl_model.py:233-416 (forward) for the
prefill path: ViT provider -> embedding(+scatter) -> RoPE -> 48 decoder layers -> logits-masked head.

Decoder layer = Megatron TransformerLayer built from the TE spec
(M/core/models/gpt/gpt_layer_specs.py:35-49): RMSNorm -> linear_qkv(+bias) -> RoPE -> core attention
-> linear_proj -> residual;  RMSNorm -> linear_fc1 -> SwiGLU -> linear_fc2 -> residual.
Weights are in MEGATRON layout: linear_qkv rows per kv group [q x (np/ng), k, v]
(R/tools/hf2mcore_long_vita.py:597-609), linear_fc1 = cat[gate, up] (:612).

Every arithmetic step is a libvita_hip.so kernel; activations live in a per-model workspace that
is reused across the 48 layers (sized once for the local sequence length).
"""
from __future__ import annotations

import os
from dataclasses import dataclass
from typing import Optional

import torch

from . import ops, parallel_state as mpu
from .dot_product_attention import DotProductAttention
from .language_model_embedding import LanguageModelEmbedding
from .layers import ColumnParallelLinear
from .rotary_pos_embedding import RotaryEmbedding


@dataclass
class GPTConfig:
    """stage3 .sh:155-200 / config_14B.json:31-56."""
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


class GPTVLModel:
    def __init__(self, cfg: GPTConfig, params: dict, external_feature_model=None):
        self.cfg, self.p = cfg, params
        self.external_feature_model = external_feature_model
        self.embedding = LanguageModelEmbedding(params["embed"])
        self.rotary_pos_emb = RotaryEmbedding(cfg.head_dim, rotary_base=cfg.rope_theta, device=params["embed"].device)
        self.core_attention = DotProductAttention(cfg.heads, cfg.kv_groups, cfg.head_dim, causal=True)
        self.output_layer = ColumnParallelLinear(params["lm_head"], bias=None)
        self._ws = {}
        # K/V all-gather messages per layer (split by kv head; gather j+1 overlaps attention j)
        self.kv_split = 4 if cfg.kv_groups % 4 == 0 else (2 if cfg.kv_groups % 2 == 0 else 1)
        self.force_cp_path = bool(int(os.environ.get("VITA_FORCE_CP", "0")))   # diagnostics only
        self.attn_events = None      # bench.py: list collecting (start, end) HIP events per attention launch

    # ---------------------------------------------------------------------------------------------
    @classmethod
    def from_oracle_layout(cls, cfg: GPTConfig, p: dict, external_feature_model=None, device="cuda"):
        def d(t):
            return t.to(device=device, dtype=torch.bfloat16).contiguous()
        q = {"embed": d(p["embed"]), "final_ln": d(p["final_ln"]), "lm_head": d(p["lm_head"]),
             "layers": [{k: d(v) for k, v in lp.items()} for lp in p["layers"]]}
        return cls(cfg, q, external_feature_model)

    @classmethod
    def random_init(cls, cfg: GPTConfig, seed: int = 1234, device="cuda", std: float = 0.02,
                    external_feature_model=None):
        """Seeded synthetic weights generated on the device (identical on every rank)."""
        g = torch.Generator(device=device).manual_seed(seed)

        def rn(*shape):
            return (torch.randn(*shape, generator=g, device=device) * std).to(torch.bfloat16)

        def ones(n):
            return torch.ones(n, dtype=torch.bfloat16, device=device)

        p = {"embed": rn(cfg.vocab, cfg.hidden), "final_ln": ones(cfg.hidden), "lm_head": rn(cfg.vocab, cfg.hidden),
             "layers": []}
        for _ in range(cfg.num_layers):
            p["layers"].append({"ln1": ones(cfg.hidden), "qkv_w": rn(cfg.qkv_out, cfg.hidden), "qkv_b": rn(cfg.qkv_out),
                                "o_w": rn(cfg.hidden, cfg.heads * cfg.head_dim), "ln2": ones(cfg.hidden),
                                "fc1_w": rn(2 * cfg.ffn, cfg.hidden), "fc2_w": rn(cfg.hidden, cfg.ffn)})
        return cls(cfg, p, external_feature_model)

    # ---------------------------------------------------------------------------------------------
    def _workspace(self, s: int, device):
        ws = self._ws.get(s)
        if ws is None:
            c = self.cfg
            e = lambda *shape: torch.empty(*shape, dtype=torch.bfloat16, device=device)  # noqa: E731
            ws = {"x": e(s, c.hidden), "qkv": e(s, c.qkv_out), "ctx": e(1, s, c.heads, c.head_dim),
                  "act": e(s, c.ffn), "kv": e(self.kv_split, 2, s, c.kv_groups // self.kv_split, c.head_dim)}
            self._ws = {s: ws}          # keep one size only
        return ws

    def decoder_layer(self, h: torch.Tensor, lp: dict, cos, sin, ws) -> torch.Tensor:
        """h [s, hidden] updated in place."""
        c = self.cfg
        s = h.shape[0]
        cp = mpu.get_context_parallel_world_size()
        use_cp = cp > 1 or self.force_cp_path          # force: exercise pack + all-gather + chunk tables at CP = 1
        x = ops.rmsnorm(h, lp["ln1"], c.eps, out=ws["x"])
        qkv = ops.gemm(x, lp["qkv_w"], ops.EPI_BIAS, lp["qkv_b"], out=ws["qkv"])
        ops.rope_qkv_(qkv, c.kv_groups, c.qpg, c.head_dim, cos, sin, ws["kv"] if use_cp else None, self.kv_split)
        m5 = qkv.view(1, s, c.kv_groups, c.qpg + 2, c.head_dim)
        q5 = m5[:, :, :, : c.qpg]                                  # grouped query view, read in place
        ev = None
        if self.attn_events is not None:
            ev = (torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True))
        if use_cp:
            ctx = self.core_attention.forward_cp(q5, ws["kv"], out=ws["ctx"], events=ev)
        else:
            if ev:
                ev[0].record()
            ctx = ops.flash_attn(q5, m5[:, :, :, c.qpg], m5[:, :, :, c.qpg + 1], causal=True, out=ws["ctx"])
            if ev:
                ev[1].record()
        if ev:
            self.attn_events.append(ev)
        ops.gemm(ctx.view(s, c.heads * c.head_dim), lp["o_w"], ops.EPI_RESIDUAL, residual=h, out=h)
        x = ops.rmsnorm(h, lp["ln2"], c.eps, out=ws["x"])
        act = ops.gemm(x, lp["fc1_w"], ops.EPI_SWIGLU, out=ws["act"])
        ops.gemm(act, lp["fc2_w"], ops.EPI_RESIDUAL, residual=h, out=h)
        return h

    # ---------------------------------------------------------------------------------------------
    def forward(self, input_ids: torch.Tensor, position_ids: Optional[torch.Tensor] = None, attention_mask=None,
                decoder_input: Optional[torch.Tensor] = None, labels=None, inference_params=None,
                packed_seq_params=None, extra_block_kwargs=None, external_inputs: Optional[dict] = None,
                tokentype_ids=None, logit_mask: Optional[torch.Tensor] = None) -> torch.Tensor:
        """Returns logits [b, n_sel (or s), vocab] (labels=None branch, gpt_vl_model.py:357-370)."""
        if labels is not None:
            raise NotImplementedError("loss / backward path is not built yet (SURVEY.md §7 step 6)")
        assert packed_seq_params is None
        if decoder_input is None:                                                         # :252-277
            if external_inputs:
                feats = self.external_feature_model(**external_inputs)                   # :267
                efd = {"features": feats}
                for k in external_inputs:
                    if "indices" in k or k == "pre_len":
                        efd[k] = external_inputs[k]
                decoder_input = self.embedding(input_ids=input_ids, position_ids=position_ids,
                                               external_feature_dict=efd)
            else:
                decoder_input = self.embedding(input_ids=input_ids, position_ids=position_ids)
        s, b, hdim = decoder_input.shape
        if b != 1:
            raise ValueError("the Long-VITA prefill path runs batch 1")
        h = decoder_input.view(s, hdim)
        rotary_seq_len = RotaryEmbedding.get_rotary_seq_len(s)                           # :289-293
        cos, sin = self.rotary_pos_emb(rotary_seq_len)                                   # :295
        ws = self._workspace(s, h.device)
        for lp in self.p["layers"]:                                                       # self.decoder(...) :299
            self.decoder_layer(h, lp, cos, sin, ws)
        # final RMSNorm is per-row, so norm only the rows the masked head keeps
        if logit_mask is not None:
            idx = ops.mask_to_index(logit_mask.transpose(0, 1).reshape(-1))
            rows = ops.row_gather(h, idx)
            sel_mask = None
        else:
            rows, sel_mask = h, None
        rows = ops.rmsnorm(rows, self.p["final_ln"], self.cfg.eps)
        logits, _ = self.output_layer(rows.view(rows.shape[0], 1, hdim), weight=None, logit_mask=sel_mask)   # :339
        if bool(torch.isnan(logits.float().sum())):                                       # :393-396
            raise ValueError("found NaN in local forward logits calculation")
        return logits.transpose(0, 1).contiguous()                                        # [s b v] -> [b s v] :370

    __call__ = forward
