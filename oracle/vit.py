"""InternViT-300M + pixel-shuffle projector restated in Megatron weight layout
(SURVEY.md §8a rows a3-a6).  TEST INFRASTRUCTURE ONLY.

Arithmetic follows the dtype of the tensors handed in: fp32 tensors give the exact-math
reference, bf16 tensors give the per-op-rounded chain the GPU path of the reference runs
(GEMMs accumulate in fp32 and round once, elementwise ops round per op)."""
from __future__ import annotations

import math
from dataclasses import dataclass

import torch
import torch.nn.functional as F

from .attention import core_attention
from .glue import pixel_shuffle


@dataclass
class ViTConfig:
    """M/pretrain_long_vita.py:183-223 (get_vision_model_args_intern_300m) + config_14B.json:2-30."""
    num_layers: int = 24
    hidden: int = 1024
    heads: int = 16
    head_dim: int = 64
    ffn: int = 4096
    patch: int = 14
    image: int = 448
    ln_eps: float = 1e-6        # block norms, SURVEY.md §9 quirk 4
    proj_ln_eps: float = 1e-5   # bare torch.nn.LayerNorm(4096)
    llm_hidden: int = 5120
    add_class_token: bool = True
    # SigLIP-400M (get_vision_model_args_siglip_400m, M/pretrain_long_vita.py:268-307; layer
    # M/core/models/vision/siglip_vit_model.py:18-86 with the local spec vit_layer_specs.py:30-53):
    activation: str = "gelu"        # "gelu_tanh": partial(F.gelu, approximate="tanh")
    layerscale: bool = True         # InternViT ls1 / ls2
    unfused_bias: bool = False      # linear_proj / linear_fc1 / linear_fc2 hand their bias back (skip_bias_add): it is
                                    # added to the bf16-rounded product in a separate op

    @classmethod
    def siglip_400m(cls, **kw):
        args = dict(num_layers=27, hidden=1152, heads=16, head_dim=72, ffn=4304, add_class_token=False,
                    activation="gelu_tanh", layerscale=False, unfused_bias=True)
        args.update(kw)
        return cls(**args)

    @property
    def grid(self):
        return self.image // self.patch

    @property
    def seq(self):
        return self.grid ** 2 + (1 if self.add_class_token else 0)


def linear(x, w, b=None):
    """fp32-accumulated GEMM rounded once to x.dtype (torch.matmul on GPU / TE linear)."""
    y = x.float() @ w.float().t()
    if b is not None:
        y = y + b.float()
    return y.to(x.dtype)


def init_vit_params(cfg: ViTConfig, seed: int = 1234, dtype=torch.bfloat16):
    """Seeded synthetic weights in MEGATRON layout: qkv rows per head [q_h, k_h, v_h]
    (L/ckpt_converter_intern_vit.py:54-66).  Linears N(0, 0.02), norms 1/0, LayerScale 0.1."""
    g = torch.Generator().manual_seed(seed)

    def rn(*shape, std=0.02):
        return (torch.randn(*shape, generator=g) * std).to(dtype)

    hd = cfg.heads * cfg.head_dim
    p = {
        "conv_w": rn(cfg.hidden, 3, cfg.patch, cfg.patch), "conv_b": rn(cfg.hidden),
        "cls": rn(1, 1, cfg.hidden, std=1.0), "pos": rn(cfg.seq, cfg.hidden, std=0.02),
        "layers": [],
        "proj_ln_w": torch.ones(4 * cfg.hidden, dtype=dtype), "proj_ln_b": torch.zeros(4 * cfg.hidden, dtype=dtype),
        "proj_fc1": rn(cfg.hidden, 4 * cfg.hidden), "proj_fc2": rn(cfg.llm_hidden, cfg.hidden),
    }
    if not cfg.add_class_token:
        del p["cls"]
    for _ in range(cfg.num_layers):
        lp = {
            "ln1_w": torch.ones(cfg.hidden, dtype=dtype), "ln1_b": torch.zeros(cfg.hidden, dtype=dtype),
            "qkv_w": rn(3 * hd, cfg.hidden), "qkv_b": rn(3 * hd),
            "proj_w": rn(cfg.hidden, hd), "proj_b": rn(cfg.hidden),
            "ls1": torch.full((cfg.hidden,), 0.1, dtype=dtype),
            "ln2_w": torch.ones(cfg.hidden, dtype=dtype), "ln2_b": torch.zeros(cfg.hidden, dtype=dtype),
            "fc1_w": rn(cfg.ffn, cfg.hidden), "fc1_b": rn(cfg.ffn),
            "fc2_w": rn(cfg.hidden, cfg.ffn), "fc2_b": rn(cfg.hidden),
            "ls2": torch.full((cfg.hidden,), 0.1, dtype=dtype),
        }
        if not cfg.layerscale:
            del lp["ls1"], lp["ls2"]
        p["layers"].append(lp)
    return p


def megatron_qkv_to_hf(w_or_b: torch.Tensor, heads: int, head_dim: int) -> torch.Tensor:
    """Inverse of the converter permutation L/ckpt_converter_intern_vit.py:54-66,100-107:
    Megatron rows [h][q|k|v][d]  ->  HF rows [q|k|v][h][d]."""
    rest = w_or_b.shape[1:]
    return w_or_b.view(heads, 3, head_dim, *rest).transpose(0, 1).reshape(3 * heads * head_dim, *rest)


def to_hf_state_dict(p, cfg: ViTConfig, prefix: str = "", projector_prefix: str = None):
    """Inverse of the converter (L/ckpt_converter_intern_vit.py:54-107): Megatron-layout ViT params -> the names of the reference's HF
    InternVisionModel (H/models/long_vita_qwen2_intern/modeling_intern_vit.py) and, with projector_prefix, of its ResamplerProjector
    (resampler_projector.py:15-22)."""
    sd = {prefix + "embeddings.class_embedding": p["cls"].reshape(1, 1, -1), prefix + "embeddings.patch_embedding.weight": p["conv_w"],
          prefix + "embeddings.patch_embedding.bias": p["conv_b"], prefix + "embeddings.position_embedding": p["pos"][None]}
    names = {"proj_w": "attn.proj.weight", "proj_b": "attn.proj.bias", "fc1_w": "mlp.fc1.weight", "fc1_b": "mlp.fc1.bias",
             "fc2_w": "mlp.fc2.weight", "fc2_b": "mlp.fc2.bias", "ln1_w": "norm1.weight", "ln1_b": "norm1.bias", "ln2_w": "norm2.weight",
             "ln2_b": "norm2.bias", "ls1": "ls1", "ls2": "ls2"}
    for i, lp in enumerate(p["layers"]):
        pre = f"{prefix}encoder.layers.{i}."
        sd[pre + "attn.qkv.weight"] = megatron_qkv_to_hf(lp["qkv_w"], cfg.heads, cfg.head_dim)
        sd[pre + "attn.qkv.bias"] = megatron_qkv_to_hf(lp["qkv_b"], cfg.heads, cfg.head_dim)
        for k, n in names.items():
            sd[pre + n] = lp[k]
    if projector_prefix is not None:
        sd.update({projector_prefix + "pre_proj_layernorm.weight": p["proj_ln_w"], projector_prefix + "pre_proj_layernorm.bias": p["proj_ln_b"],
                   projector_prefix + "mlp.0.weight": p["proj_fc1"], projector_prefix + "mlp.2.weight": p["proj_fc2"]})
    return sd


def vit_embed(images, p, cfg: ViTConfig):
    """M/core/models/vision/intern_vit_model.py:203-216: conv14/14 + cls + learned pos-emb -> [b, s, h]."""
    x = F.conv2d(images.float(), p["conv_w"].float(), p["conv_b"].float(), stride=cfg.patch).to(images.dtype)
    x = x.reshape(x.shape[0], x.shape[1], -1).permute(0, 2, 1)
    if cfg.add_class_token:
        x = torch.cat([p["cls"].expand(x.shape[0], -1, -1).to(x.dtype), x], dim=1)
    return (x + p["pos"][None].to(x.dtype)).contiguous()        # layout only (the reference makes it contiguous at :243-244)


def _act(y, cfg: ViTConfig):
    if cfg.activation == "gelu_tanh":
        return F.gelu(y.float(), approximate="tanh").to(y.dtype)
    return F.gelu(y.float()).to(y.dtype)


def _linear_bias(x, w, b, cfg: ViTConfig):
    """TE linear (InternViT spec): bias inside the fp32 epilogue; Megatron local spec with skip_bias_add (SigLIP):
    the product is rounded first and the bias added by a separate bf16 op."""
    if cfg.unfused_bias:
        return linear(x, w) + b.to(x.dtype)
    return linear(x, w, b)


def vit_layer(x, lp, cfg: ViTConfig):
    """InternViTTransformerLayer.forward, intern_vit_model.py:32-89 / SigLIPViTTransformerLayer.forward,
    siglip_vit_model.py:29-86; x [b, s, h].  QKV split per head [q, k, v] (Megatron SelfAttention with ng == np)."""
    b, s, h = x.shape
    res = x
    y = F.layer_norm(x.float(), (h,), lp["ln1_w"].float(), lp["ln1_b"].float(), cfg.ln_eps).to(x.dtype)
    qkv = linear(y, lp["qkv_w"], lp["qkv_b"]).view(b, s, cfg.heads, 3 * cfg.head_dim)
    q, k, v = torch.split(qkv, cfg.head_dim, dim=-1)
    ctx = core_attention(q.transpose(0, 1), k.transpose(0, 1), v.transpose(0, 1), causal=False)  # [s, b, h]
    out = _linear_bias(ctx.transpose(0, 1), lp["proj_w"], lp["proj_b"], cfg)
    x = res + (out * lp["ls1"].to(x.dtype) if cfg.layerscale else out)
    res = x
    y = F.layer_norm(x.float(), (h,), lp["ln2_w"].float(), lp["ln2_b"].float(), cfg.ln_eps).to(x.dtype)
    y = _act(_linear_bias(y, lp["fc1_w"], lp["fc1_b"], cfg), cfg)
    y = _linear_bias(y, lp["fc2_w"], lp["fc2_b"], cfg)
    return res + (y * lp["ls2"].to(x.dtype) if cfg.layerscale else y)


def vit_project(x, p, cfg: ViTConfig):
    """forward_downsample + forward_projection, M/pretrain_long_vita.py:452-483."""
    if cfg.add_class_token:
        x = x[:, 1:, :]
    n = x.shape[0]
    x = x.reshape(n, cfg.grid, cfg.grid, -1)
    x = pixel_shuffle(x, 0.5)
    x = x.reshape(n, -1, x.shape[-1])
    y = F.layer_norm(x.float(), (x.shape[-1],), p["proj_ln_w"].float(), p["proj_ln_b"].float(), cfg.proj_ln_eps).to(x.dtype)
    y = linear(y, p["proj_fc1"])
    y = F.gelu(y.float()).to(x.dtype)
    return linear(y, p["proj_fc2"])


def vision_model(images, p, cfg: ViTConfig):
    """MegatronVisionModel.forward_once (M/pretrain_long_vita.py:485-520): [N,3,H,W] -> [N,256,5120]."""
    x = vit_embed(images, p, cfg)
    for lp in p["layers"]:
        x = vit_layer(x, lp, cfg)
    return vit_project(x, p, cfg)
