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
    # SigLIP-400M (get_vision_model_args_siglip_400m, :268-307): tanh GELU, no LayerScale, no class token, biases of
    # linear_proj / fc1 / fc2 added unfused (Megatron local layer spec, vit_layer_specs.py:30-53)
    activation: str = "gelu"
    layerscale: bool = True
    unfused_bias: bool = False

    @classmethod
    def siglip_400m(cls, **kw):
        args = dict(num_layers=27, hidden=1152, heads=16, head_dim=72, ffn=4304, add_class_token=False,
                    activation="gelu_tanh", layerscale=False, unfused_bias=True)
        args.update(kw)
        return cls(**args)

    @property
    def head_dim_pad(self):            # the flash kernel runs d = 64 / 96 / 128: other head sizes are zero-padded at load time
        return min(d for d in (64, 96, 128) if d >= self.head_dim)

    @property
    def ffn_pad(self):                 # fc2's K must be a multiple of the GEMM's BK = 64
        return -(-self.ffn // 64) * 64

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
        """p: dict produced by oracle.vit.init_vit_params / checkpoint.hf_vit_to_params (plain tensors, Megatron layout).
        Head size and FFN width are zero-padded to what the kernels tile (72 -> 96 [r05; 128 before], 4304 -> 4352 for SigLIP): padded q / k
        columns contribute 0 to the scores, padded v columns produce 0 context that meets zero proj_w columns, padded
        fc1 rows give gelu(0) = 0 that meets zero fc2_w columns — the function is unchanged."""
        def d(t):
            return t.to(device=device, dtype=torch.bfloat16).contiguous()
        h, nh, hd, hp, f, fp = cfg.hidden, cfg.heads, cfg.head_dim, cfg.head_dim_pad, cfg.ffn, cfg.ffn_pad
        conv = torch.zeros(h, cls.K_PAD, dtype=torch.bfloat16)
        conv[:, :588] = p["conv_w"].reshape(h, 588).to(torch.bfloat16)
        q = {"conv_w": d(conv), "conv_b": d(p["conv_b"]), "pos": d(p["pos"]),
             "proj_ln_w": d(p["proj_ln_w"]), "proj_ln_b": d(p["proj_ln_b"]), "proj_fc1": d(p["proj_fc1"]),
             "proj_fc2": d(p["proj_fc2"]), "layers": []}
        if cfg.add_class_token:
            q["cls"] = d(p["cls"].reshape(-1))
        for lp in p["layers"]:
            o = {k: lp[k] for k in ("ln1_w", "ln1_b", "ln2_w", "ln2_b", "proj_b", "fc2_b")}
            o["ls1"] = lp["ls1"] if cfg.layerscale else torch.ones(h)
            o["ls2"] = lp["ls2"] if cfg.layerscale else torch.ones(h)
            qkv_w = torch.zeros(nh, 3, hp, h, dtype=lp["qkv_w"].dtype)
            qkv_w[:, :, :hd] = lp["qkv_w"].view(nh, 3, hd, h)
            qkv_b = torch.zeros(nh, 3, hp, dtype=lp["qkv_b"].dtype)
            qkv_b[:, :, :hd] = lp["qkv_b"].view(nh, 3, hd)
            proj_w = torch.zeros(h, nh, hp, dtype=lp["proj_w"].dtype)
            proj_w[:, :, :hd] = lp["proj_w"].view(h, nh, hd)
            fc1_w = torch.zeros(fp, h, dtype=lp["fc1_w"].dtype)
            fc1_w[:f] = lp["fc1_w"]
            fc1_b = torch.zeros(fp, dtype=lp["fc1_b"].dtype)
            fc1_b[:f] = lp["fc1_b"]
            fc2_w = torch.zeros(h, fp, dtype=lp["fc2_w"].dtype)
            fc2_w[:, :f] = lp["fc2_w"]
            o.update(qkv_w=qkv_w.view(nh * 3 * hp, h), qkv_b=qkv_b.view(-1), proj_w=proj_w.view(h, nh * hp), fc1_w=fc1_w,
                     fc1_b=fc1_b, fc2_w=fc2_w)
            q["layers"].append({k: d(v) for k, v in o.items()})
        return cls(cfg, q)

    @staticmethod
    def random_params(cfg: VisionConfig, seed: int = 1234, device="cuda", std: float = 0.02) -> dict:
        """Seeded synthetic weights in the un-padded, per-tensor ("oracle") layout `from_oracle_layout` takes; generated on the device,
        returned on the host.  bench.py's tower is `from_oracle_layout(random_params(seed=4321))`: the parity tests hand the same
        dict to the oracle (tests/test_parity_bench_gpu.py)."""
        g = torch.Generator(device=device).manual_seed(seed)

        def rn(*shape, s=std):
            return (torch.randn(*shape, generator=g, device=device) * s).to(torch.bfloat16)

        def ones(n, v=1.0):
            return torch.full((n,), v, dtype=torch.bfloat16, device=device)

        h, hd_all = cfg.hidden, cfg.heads * cfg.head_dim
        seq = cfg.grid ** 2 + int(cfg.add_class_token)
        p = {"conv_w": rn(h, 3, cfg.patch, cfg.patch), "conv_b": rn(h), "cls": rn(h, s=1.0), "pos": rn(seq, h), "layers": [],
             "proj_ln_w": ones(4 * h), "proj_ln_b": ones(4 * h, 0.0), "proj_fc1": rn(h, 4 * h),
             "proj_fc2": rn(cfg.llm_hidden, h)}
        for _ in range(cfg.num_layers):
            p["layers"].append({"ln1_w": ones(h), "ln1_b": ones(h, 0.0), "qkv_w": rn(3 * hd_all, h), "qkv_b": rn(3 * hd_all),
                                "proj_w": rn(h, hd_all), "proj_b": rn(h), "ls1": ones(h, 0.1),
                                "ln2_w": ones(h), "ln2_b": ones(h, 0.0), "fc1_w": rn(cfg.ffn, h), "fc1_b": rn(cfg.ffn),
                                "fc2_w": rn(h, cfg.ffn), "fc2_b": rn(h), "ls2": ones(h, 0.1)})
        p = {k: (v.cpu() if torch.is_tensor(v) else [{kk: vv.cpu() for kk, vv in lp.items()} for lp in v]) for k, v in p.items()}
        return p

    @classmethod
    def random_init(cls, cfg: VisionConfig, seed: int = 1234, device="cuda", std: float = 0.02):
        return cls.from_oracle_layout(cfg, cls.random_params(cfg, seed, device, std), device)

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
        hp = cfg.head_dim_pad
        scale = 1.0 / (cfg.head_dim ** 0.5)                    # of the TRUE head size (padding adds zeros to q.k only)
        epi_act = ops.EPI_BIAS2_GELU_TANH if cfg.activation == "gelu_tanh" else ops.EPI_BIAS_GELU
        if cfg.unfused_bias != (cfg.activation == "gelu_tanh"):
            raise NotImplementedError("built combinations: InternViT (fused bias, erf GELU), SigLIP (unfused bias, tanh GELU)")
        for lp in p["layers"]:                    # InternViTTransformerLayer :32-89 / SigLIPViTTransformerLayer :29-86
            y = ops.layernorm(x2, lp["ln1_w"], lp["ln1_b"], cfg.ln_eps)
            qkv = ops.gemm(y, lp["qkv_w"], ops.EPI_BIAS, lp["qkv_b"]).view(n, seq, cfg.heads, 3, hp)
            ctx = ops.flash_attn(qkv[:, :, :, 0], qkv[:, :, :, 1], qkv[:, :, :, 2], causal=False, softmax_scale=scale)
            ctx2 = ctx.view(n * seq, cfg.heads * hp)
            if cfg.unfused_bias:
                ops.gemm(ctx2, lp["proj_w"], ops.EPI_BIAS2_RES, lp["proj_b"], residual=x2, out=x2)
            else:
                ops.gemm(ctx2, lp["proj_w"], ops.EPI_BIAS_SCALE_RES, lp["proj_b"], lp["ls1"], x2, out=x2)
            y = ops.layernorm(x2, lp["ln2_w"], lp["ln2_b"], cfg.ln_eps, out=y)
            f = ops.gemm(y, lp["fc1_w"], epi_act, lp["fc1_b"])
            if cfg.unfused_bias:
                ops.gemm(f, lp["fc2_w"], ops.EPI_BIAS2_RES, lp["fc2_b"], residual=x2, out=x2)
            else:
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
