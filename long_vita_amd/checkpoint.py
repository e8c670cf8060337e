"""Weight-layout converters and checkpoint readers into the HIP modules.

The modules of this package keep MEGATRON's fused layouts (gpt_vl_model.py / vision.py), so

  * a Megatron-core checkpoint (`Long-VITA-*_MG`: iter_XXXXXXX/mp_rank_YY/model_optim_rng.pt, one file per
    tensor-parallel rank) only needs its TP shards merged — `merge_tp_shards`, the inverse of the chunking in
    L/ckpt_converter_intern_vit.py:150-158 and of Megatron's Column/RowParallelLinear partitioning — and a rename
    (`mcore_llm_to_params`, `mcore_vit_to_params`);
  * a transformers checkpoint (`Long-VITA-*_HF`: *.safetensors) needs the re-layout of
    R/tools/hf2mcore_long_vita.py:597-613 (LLM: q/k/v -> per-kv-group [q x (np/ng), k, v]; gate/up -> fc1) and of
    L/ckpt_converter_intern_vit.py:54-66,100-107 (ViT: [q|k|v][head] -> [head][q|k|v]) — `hf_llm_to_params`,
    `hf_vit_to_params`.

Everything here is tensor slicing / concatenation on the host (no arithmetic); `GPTVLModel.from_oracle_layout` /
`MegatronVisionModel.from_oracle_layout` then move the dicts to the device in bf16.
"""
from __future__ import annotations

import glob
import json
import os
import re
from typing import Dict, List, Optional

import torch

# ------------------------------------------------------------------------------------------------
# transformers (HF) names
# ------------------------------------------------------------------------------------------------
def hf_llm_to_params(sd: Dict[str, torch.Tensor], cfg, prefix: str = "") -> dict:
    """Qwen2ForCausalLM names -> {"embed", "final_ln", "lm_head", "layers": [{ln1, qkv_w, qkv_b, o_w, ln2, fc1_w, fc2_w}]}
    (R/tools/hf2mcore_long_vita.py:590-617).  cfg: GPTConfig / oracle LLMConfig (hidden, heads, kv_groups, head_dim)."""
    d, ng, h = cfg.head_dim, cfg.kv_groups, cfg.hidden

    def g(name):
        return sd[prefix + name]

    embed = g("model.embed_tokens.weight")
    p = {"embed": embed, "final_ln": g("model.norm.weight"),
         "lm_head": sd.get(prefix + "lm_head.weight", embed),          # tied embeddings: lm_head absent
         "layers": []}
    for i in range(cfg.num_layers):
        pre = f"model.layers.{i}."
        q = g(pre + "self_attn.q_proj.weight").view(ng, -1, d, h)       # :597-600
        k = g(pre + "self_attn.k_proj.weight").view(ng, -1, d, h)
        v = g(pre + "self_attn.v_proj.weight").view(ng, -1, d, h)
        qkv_w = torch.cat([q, k, v], dim=1).reshape(-1, h).contiguous()
        qb = g(pre + "self_attn.q_proj.bias").view(ng, -1)              # :603-606
        kb = g(pre + "self_attn.k_proj.bias").view(ng, -1)
        vb = g(pre + "self_attn.v_proj.bias").view(ng, -1)
        qkv_b = torch.cat([qb, kb, vb], dim=1).reshape(-1).contiguous()
        fc1 = torch.cat([g(pre + "mlp.gate_proj.weight"), g(pre + "mlp.up_proj.weight")])       # :610
        p["layers"].append({"ln1": g(pre + "input_layernorm.weight"), "qkv_w": qkv_w, "qkv_b": qkv_b,
                            "o_w": g(pre + "self_attn.o_proj.weight"), "ln2": g(pre + "post_attention_layernorm.weight"),
                            "fc1_w": fc1, "fc2_w": g(pre + "mlp.down_proj.weight")})
    return p


def hf_qkv_to_megatron(w_or_b: torch.Tensor, heads: int, head_dim: int) -> torch.Tensor:
    """InternViT attn.qkv rows [q|k|v][head][d] -> Megatron rows [head][q|k|v][d]
    (the `indices` gather of L/ckpt_converter_intern_vit.py:54-66,100-107)."""
    rest = w_or_b.shape[1:]
    return w_or_b.reshape(3, heads, head_dim, *rest).transpose(0, 1).reshape(3 * heads * head_dim, *rest).contiguous()


def hf_vit_to_params(sd: Dict[str, torch.Tensor], vcfg, prefix: str = "", projector_prefix: Optional[str] = None) -> dict:
    """InternVisionModel names (+ ResamplerProjector: pre_proj_layernorm, mlp.0, mlp.2 — bias-free,
    H/models/long_vita_qwen2_intern/resampler_projector.py:15-22) -> the vision params of vision.py."""
    def g(name):
        return sd[prefix + name]

    p = {"conv_w": g("embeddings.patch_embedding.weight"), "conv_b": g("embeddings.patch_embedding.bias"),
         "cls": g("embeddings.class_embedding").reshape(1, 1, -1),                     # :80-83
         "pos": g("embeddings.position_embedding").squeeze(0), "layers": []}            # :84-86
    for i in range(vcfg.num_layers):
        pre = f"encoder.layers.{i}."
        p["layers"].append({
            "ln1_w": g(pre + "norm1.weight"), "ln1_b": g(pre + "norm1.bias"),
            "qkv_w": hf_qkv_to_megatron(g(pre + "attn.qkv.weight"), vcfg.heads, vcfg.head_dim),
            "qkv_b": hf_qkv_to_megatron(g(pre + "attn.qkv.bias"), vcfg.heads, vcfg.head_dim),
            "proj_w": g(pre + "attn.proj.weight"), "proj_b": g(pre + "attn.proj.bias"), "ls1": g(pre + "ls1"),
            "ln2_w": g(pre + "norm2.weight"), "ln2_b": g(pre + "norm2.bias"),
            "fc1_w": g(pre + "mlp.fc1.weight"), "fc1_b": g(pre + "mlp.fc1.bias"),
            "fc2_w": g(pre + "mlp.fc2.weight"), "fc2_b": g(pre + "mlp.fc2.bias"), "ls2": g(pre + "ls2")})
    if projector_prefix is not None:
        q = projector_prefix
        p.update({"proj_ln_w": sd[q + "pre_proj_layernorm.weight"], "proj_ln_b": sd[q + "pre_proj_layernorm.bias"],
                  "proj_fc1": sd[q + "mlp.0.weight"], "proj_fc2": sd[q + "mlp.2.weight"]})
    return p


# ------------------------------------------------------------------------------------------------
# Megatron-core names
# ------------------------------------------------------------------------------------------------
_COLUMN = ("linear_qkv.weight", "linear_qkv.bias", "linear_fc1.weight", "linear_fc1.bias", "word_embeddings.weight",
           "output_layer.weight", "q_layernorm.weight", "q_layernorm.bias", "k_layernorm.weight", "k_layernorm.bias")
_ROW = ("linear_proj.weight", "linear_fc2.weight")


_VISION_MARKS = ("external_feature_model.", "vision_model", "vision_projection.")


def _is_vision_key(name: str) -> bool:
    """Keys of the ViT / projector inside a combined checkpoint (`external_feature_model.vit.*`,
    `external_feature_model.vision_projection.*`: M/pretrain_long_vita.py:381,436) or after M/ckpt_split_llm_and_vit.py
    stripped that prefix (`vit.*`, `vision_projection.*`).  Their MLPs are GELU: linear_fc1 is NOT a [gate; up] pair."""
    return name.startswith("vit.") or any(m in name for m in _VISION_MARKS)


def merge_tp_shards(shards: List[Dict[str, torch.Tensor]], swiglu_fc1: bool = True) -> Dict[str, torch.Tensor]:
    """Inverse of Megatron's tensor-parallel partitioning: column-parallel tensors (and the vocab-parallel embedding /
    output layer) are concatenated on dim 0, row-parallel weights on dim 1, everything else is replicated.  A SwiGLU
    linear_fc1 shard is [gate_shard; up_shard], so its halves are merged separately — decided per key: only the
    decoder's (LLM) linear_fc1 is gated; the ViT's and the projector's fc1 (GELU) are plain column-parallel, whatever
    prefix they carry.  swiglu_fc1=False: no key is gated (the stand-alone ViT checkpoint, whose layers are un-prefixed)."""
    if len(shards) == 1:
        return dict(shards[0])
    out = {}
    for name, t0 in shards[0].items():
        if t0 is None or not isinstance(t0, torch.Tensor):                    # TE _extra_state
            continue
        parts = [s[name] for s in shards]
        if any(name.endswith(sfx) for sfx in _COLUMN):
            if swiglu_fc1 and name.endswith("linear_fc1.weight") and not _is_vision_key(name):
                halves = [p.chunk(2, dim=0) for p in parts]
                out[name] = torch.cat([h[0] for h in halves] + [h[1] for h in halves], dim=0)
            else:
                out[name] = torch.cat(parts, dim=0)
        elif any(name.endswith(sfx) for sfx in _ROW):
            out[name] = torch.cat(parts, dim=1)
        else:
            out[name] = t0
    return out


def mcore_llm_to_params(sd: Dict[str, torch.Tensor], cfg, prefix: str = "") -> dict:
    """Megatron-core GPTModel names with the TE layer spec (layer norms folded into the following linear,
    M/core/models/gpt/gpt_layer_specs.py:35-49; names as in R/tools/hf2mcore_long_vita.py:590-617)."""
    def g(*names):
        for name in names:                       # TE-spec name first, then the local-spec name
            if prefix + name in sd:              # (M/ckpt_convert_modellink_to_megatron_with_te.py:37-41 renames one into the other)
                return sd[prefix + name]
        raise KeyError(prefix + names[0])

    embed = g("embedding.word_embeddings.weight")
    p = {"embed": embed, "final_ln": g("decoder.final_layernorm.weight"),
         "lm_head": sd.get(prefix + "output_layer.weight", embed), "layers": []}
    for i in range(cfg.num_layers):
        pre = f"decoder.layers.{i}."
        p["layers"].append({"ln1": g(pre + "self_attention.linear_qkv.layer_norm_weight", pre + "input_layernorm.weight"),
                            "qkv_w": g(pre + "self_attention.linear_qkv.weight"),
                            "qkv_b": g(pre + "self_attention.linear_qkv.bias"),
                            "o_w": g(pre + "self_attention.linear_proj.weight"),
                            "ln2": g(pre + "mlp.linear_fc1.layer_norm_weight", pre + "pre_mlp_layernorm.weight"),
                            "fc1_w": g(pre + "mlp.linear_fc1.weight"), "fc2_w": g(pre + "mlp.linear_fc2.weight")})
    # the vocabulary is padded to a multiple of 128 * TP on the Megatron side (--make-vocab-size-divisible-by)
    return p


def mcore_vit_to_params(sd: Dict[str, torch.Tensor], vcfg, prefix: str = "") -> dict:
    """Names written by L/ckpt_converter_intern_vit.py:76-141 (with --use-te, or the local-spec names without it)."""
    def g(*names):
        for name in names:
            if prefix + name in sd:
                return sd[prefix + name]
        raise KeyError(prefix + names[0])

    p = {"conv_w": g("conv1.weight"), "conv_b": g("conv1.bias"), "cls": g("class_token").reshape(1, 1, -1),
         "pos": g("position_embeddings.weight"), "layers": []}
    for i in range(vcfg.num_layers):
        pre = f"decoder.layers.{i}."
        p["layers"].append({
            "ln1_w": g(pre + "self_attention.linear_qkv.layer_norm_weight", pre + "input_layernorm.weight"),
            "ln1_b": g(pre + "self_attention.linear_qkv.layer_norm_bias", pre + "input_layernorm.bias"),
            "qkv_w": g(pre + "self_attention.linear_qkv.weight"), "qkv_b": g(pre + "self_attention.linear_qkv.bias"),
            "proj_w": g(pre + "self_attention.linear_proj.weight"), "proj_b": g(pre + "self_attention.linear_proj.bias"),
            "ls1": g(pre + "ls1"),
            "ln2_w": g(pre + "mlp.linear_fc1.layer_norm_weight", pre + "pre_mlp_layernorm.weight"),
            "ln2_b": g(pre + "mlp.linear_fc1.layer_norm_bias", pre + "pre_mlp_layernorm.bias"),
            "fc1_w": g(pre + "mlp.linear_fc1.weight"), "fc1_b": g(pre + "mlp.linear_fc1.bias"),
            "fc2_w": g(pre + "mlp.linear_fc2.weight"), "fc2_b": g(pre + "mlp.linear_fc2.bias"), "ls2": g(pre + "ls2")})
    return p


def split_llm_and_vit(sd: Dict[str, torch.Tensor]):
    """A combined training checkpoint -> (language-model part, vision part), as M/ckpt_split_llm_and_vit.py:21-59 does it
    per file: keys containing "unused" are dropped, keys containing "external_feature_model." go to the vision part with
    everything up to and including that prefix removed."""
    llm, vit = {}, {}
    for k, v in sd.items():
        if "unused" in k:
            continue
        if "external_feature_model." in k:
            vit[k.split("external_feature_model.")[-1]] = v
        else:
            llm[k] = v
    return llm, vit


# ------------------------------------------------------------------------------------------------
# files
# ------------------------------------------------------------------------------------------------
def load_hf_safetensors(path: str) -> Dict[str, torch.Tensor]:
    """All tensors of a transformers checkpoint directory (model.safetensors or the sharded form with
    model.safetensors.index.json), on the host."""
    from safetensors import safe_open
    index = os.path.join(path, "model.safetensors.index.json")
    if os.path.exists(index):
        files = sorted(set(json.load(open(index))["weight_map"].values()))
    else:
        files = sorted(os.path.basename(f) for f in glob.glob(os.path.join(path, "*.safetensors")))
    if not files:
        raise FileNotFoundError(f"no .safetensors files under {path}")
    sd = {}
    for f in files:
        with safe_open(os.path.join(path, f), framework="pt", device="cpu") as fh:
            for k in fh.keys():
                sd[k] = fh.get_tensor(k)
    return sd


def load_mcore_checkpoint(path: str, iteration: Optional[int] = None, swiglu_fc1: Optional[bool] = None) -> Dict[str, torch.Tensor]:
    """<path>/latest_checkpointed_iteration.txt + iter_XXXXXXX/mp_rank_YY/model_optim_rng.pt (M/training/checkpointing.py
    layout, pipeline size 1) -> merged 'model' state dict.  swiglu_fc1: are the un-prefixed linear_fc1 shards [gate; up]
    pairs?  Default: yes, unless this is the stand-alone ViT checkpoint L/ckpt_converter_intern_vit.py writes (the one
    `--vit-load` points at; top-level conv1 / class_token, GELU MLP, no `vision_model.` prefix)."""
    if iteration is None:
        tag = open(os.path.join(path, "latest_checkpointed_iteration.txt")).read().strip()
        it_dir = "release" if tag == "release" else f"iter_{int(tag):07d}"
    else:
        it_dir = f"iter_{iteration:07d}"
    dirs = os.listdir(os.path.join(path, it_dir))
    ranks = sorted(d for d in dirs if re.fullmatch(r"mp_rank_\d\d", d))
    if not ranks:                                  # Megatron appends the pipeline rank when PP > 1: mp_rank_TT_PPP
        staged = sorted(d for d in dirs if re.fullmatch(r"mp_rank_\d\d_\d\d\d", d))
        if any(not d.endswith("_000") for d in staged):
            raise NotImplementedError("pipeline-parallel checkpoints are not merged here (this path runs PP = 1)")
        ranks = staged
    if not ranks:
        raise FileNotFoundError(f"no mp_rank_XX directories under {os.path.join(path, it_dir)}")
    shards = [torch.load(os.path.join(path, it_dir, r, "model_optim_rng.pt"), map_location="cpu", weights_only=False)["model"]
              for r in ranks]
    if swiglu_fc1 is None:      # stand-alone ViT checkpoint: top-level conv1.weight, every un-prefixed layer is a ViT layer
        swiglu_fc1 = "conv1.weight" not in shards[0]
    return merge_tp_shards(shards, swiglu_fc1=swiglu_fc1)
