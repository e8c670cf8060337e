"""ViT + pixel-shuffle projector — mirror of MegatronVisionModel (M/pretrain_long_vita.py:310-596)
and InternViTModel / InternViTTransformerLayer (M/core/models/vision/intern_vit_model.py).

Weights are kept in MEGATRON layout (ViT linear_qkv rows per head [q_h, k_h, v_h],
L/ckpt_converter_intern_vit.py:54-66).  Activations are [frames, tokens, hidden] (the reference's
[s, b, h] transposes are layout conventions of Megatron's TransformerBlock, not arithmetic).
"""
from __future__ import annotations

from dataclasses import dataclass
from typing import Optional

import torch

from . import ops


@dataclass
class VisionConfig:
    """get_vision_model_args_intern_300m (M/pretrain_long_vita.py:183-223) + projector (:403-446)."""
    num_layers: int = 24
    hidden: int = 1024
    heads: int = 16
    head_dim: int = 64
    ffn: int = 4096
    patch: int = 14
    image: int = 448
    ln_eps: float = 1e-6
    proj_ln_eps: float = 1e-5
    llm_hidden: int = 5120
    add_class_token: bool = True
    chunk_frames: int = 256            # forward_chunk, :522-533

    @property
    def grid(self):
        return self.image // self.patch


class MegatronVisionModel:
    """`external_feature_model(**external_inputs)` with key `images` -> [N, 256, llm_hidden]."""

    K_PAD = 640   # 3*14*14 = 588 padded to a multiple of the GEMM K tile

    def __init__(self, cfg: VisionConfig, params: dict):
        self.cfg, self.p = cfg, params

    @classmethod
    def from_oracle_layout(cls, cfg: VisionConfig, p: dict, device="cuda"):
        """p: dict produced by oracle.vit.init_vit_params (plain tensors, Megatron layout)."""
        def d(t):
            return t.to(device=device, dtype=torch.bfloat16).contiguous()
        conv = torch.zeros(cfg.hidden, cls.K_PAD, dtype=torch.bfloat16)
        conv[:, :588] = p["conv_w"].reshape(cfg.hidden, 588).to(torch.bfloat16)
        q = {"conv_w": d(conv), "conv_b": d(p["conv_b"]), "cls": d(p["cls"].reshape(-1)), "pos": d(p["pos"]),
             "proj_ln_w": d(p["proj_ln_w"]), "proj_ln_b": d(p["proj_ln_b"]), "proj_fc1": d(p["proj_fc1"]),
             "proj_fc2": d(p["proj_fc2"]), "layers": [{k: d(v) for k, v in lp.items()} for lp in p["layers"]]}
        return cls(cfg, q)

    @classmethod
    def random_init(cls, cfg: VisionConfig, seed: int = 1234, device="cuda", std: float = 0.02):
        g = torch.Generator(device=device).manual_seed(seed)

        def rn(*shape, s=std):
            return (torch.randn(*shape, generator=g, device=device) * s).to(torch.bfloat16)

        def ones(n, v=1.0):
            return torch.full((n,), v, dtype=torch.bfloat16, device=device)

        h = cfg.hidden
        conv = rn(h, cls.K_PAD)
        conv[:, 588:] = 0
        seq = cfg.grid ** 2 + int(cfg.add_class_token)
        p = {"conv_w": conv, "conv_b": rn(h), "cls": rn(h, s=1.0), "pos": rn(seq, h), "layers": [],
             "proj_ln_w": ones(4 * h), "proj_ln_b": ones(4 * h, 0.0), "proj_fc1": rn(h, 4 * h),
             "proj_fc2": rn(cfg.llm_hidden, h)}
        for _ in range(cfg.num_layers):
            p["layers"].append({"ln1_w": ones(h), "ln1_b": ones(h, 0.0), "qkv_w": rn(3 * h, h), "qkv_b": rn(3 * h),
                                "proj_w": rn(h, h), "proj_b": rn(h), "ls1": ones(h, 0.1),
                                "ln2_w": ones(h), "ln2_b": ones(h, 0.0), "fc1_w": rn(cfg.ffn, h), "fc1_b": rn(cfg.ffn),
                                "fc2_w": rn(h, cfg.ffn), "fc2_b": rn(h), "ls2": ones(h, 0.1)})
        return cls(cfg, p)

    # -- InternViTModel.forward, intern_vit_model.py:190-261 ---------------------------------------
    def vit(self, images: torch.Tensor) -> torch.Tensor:
        cfg, p = self.cfg, self.p
        n = images.shape[0]
        npatch = cfg.grid ** 2
        patches = ops.patchify14(images, self.K_PAD)                                     # conv1 as im2col
        pe = ops.gemm(patches, p["conv_w"], ops.EPI_BIAS, p["conv_b"])                  # :203
        del patches
        x = ops.vit_assemble(pe, p["cls"] if cfg.add_class_token else None, p["pos"], n, npatch)   # :207-216
        del pe
        seq, h = x.shape[1], cfg.hidden
        x2 = x.view(n * seq, h)
        for lp in p["layers"]:                                                           # InternViTTransformerLayer :32-89
            y = ops.layernorm(x2, lp["ln1_w"], lp["ln1_b"], cfg.ln_eps)
            qkv = ops.gemm(y, lp["qkv_w"], ops.EPI_BIAS, lp["qkv_b"]).view(n, seq, cfg.heads, 3, cfg.head_dim)
            ctx = ops.flash_attn(qkv[:, :, :, 0], qkv[:, :, :, 1], qkv[:, :, :, 2], causal=False)
            ops.gemm(ctx.view(n * seq, h), lp["proj_w"], ops.EPI_BIAS_SCALE_RES, lp["proj_b"], lp["ls1"], x2, out=x2)
            y = ops.layernorm(x2, lp["ln2_w"], lp["ln2_b"], cfg.ln_eps, out=y)
            f = ops.gemm(y, lp["fc1_w"], ops.EPI_BIAS_GELU, lp["fc1_b"])
            ops.gemm(f, lp["fc2_w"], ops.EPI_BIAS_SCALE_RES, lp["fc2_b"], lp["ls2"], x2, out=x2)
            del y, qkv, ctx, f
        return x

    # -- forward_downsample + forward_projection, M/pretrain_long_vita.py:452-483 ---------------------
    def project(self, vit_output: torch.Tensor) -> torch.Tensor:
        cfg, p = self.cfg, self.p
        n = vit_output.shape[0]
        t = ops.pixel_shuffle_ln(vit_output, p["proj_ln_w"], p["proj_ln_b"], cfg.grid, cfg.add_class_token,
                                 cfg.proj_ln_eps)                                        # [n, 256, 4h]
        f = ops.gemm(t.view(-1, t.shape[-1]), p["proj_fc1"], ops.EPI_BIAS_GELU)
        o = ops.gemm(f, p["proj_fc2"])
        return o.view(n, -1, cfg.llm_hidden)

    def forward_once(self, images):
        return self.project(self.vit(images))

    def forward(self, **kw_args) -> torch.Tensor:
        images = kw_args["images"]
        outs = [self.forward_once(chunk) for chunk in torch.split(images, self.cfg.chunk_frames, dim=0)]   # forward_chunk
        return outs[0] if len(outs) == 1 else torch.cat(outs, dim=0)

    __call__ = forward
