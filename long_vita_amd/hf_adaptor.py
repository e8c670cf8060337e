"""The transformers entry point on the HIP path (VERDICT r05 "next round" item 5).

`R/tools/inference_long_vita.py:797-875` is the reference's HF-side script: `AutoModelForCausalLM.from_pretrained(model_path,
trust_remote_code=True, device_map=..., torch_dtype=torch.bfloat16, attn_implementation="flash_attention_2").eval()`, a
`model.generation_config` it edits, and `model.generate(inputs=inputs, images=images, image_indices=image_indices)`.  The class behind
that call is `LongVITAForCausalLM` (H/models/long_vita_qwen2_intern/modeling_long_vita.py:238-327) over `LongVITAModel.forward`
(:74-221): InternVisionModel -> drop the class token -> ResamplerProjector (:91-98), `inputs_embeds[indices_b, indices_s] =
image_embeds` (:137-147), the Qwen2 decoder stack, `lm_head` on the last `num_logits_to_keep` rows (:308).

This module puts the HIP kernels behind that class's call surface: the same constructor path (`from_pretrained` on a `*_HF` checkpoint
directory, or `from_state_dict` on a `LongVITAForCausalLM`-layout state dict), `forward(input_ids, attention_mask, images,
image_indices, position_ids, past_key_values, inputs_embeds, labels, use_cache, ..., num_logits_to_keep)` returning
`CausalLMOutputWithPast`, greedy / sampled `generate(inputs=, images=, image_indices=)` returning `[1, prompt + new]` token ids,
`eval()`, `generation_config`.  The script runs on it with one changed import.  Weights are re-laid out
once by `checkpoint.hf_llm_to_params` / `hf_vit_to_params` (the re-layout of R/tools/hf2mcore_long_vita.py:597-613); everything
arithmetic is `GPTVLModel` / `MegatronVisionModel`, i.e. libvita_hip.so kernels.  No torch / transformers arithmetic runs here, and
there is no fallback: without the library `lib.load` raises.

What differs from the transformers class, on purpose: batch 1 only (the Long-VITA path; a padded `attention_mask` raises);
`past_key_values` is this package's own cache object (`HipCache`: the sharded KV cache of inference_params.py), opaque to the caller
as transformers' `Cache` is; `labels` raises (training goes through pretrain_long_vita.py's path, training.py);
`output_attentions` / `output_hidden_states` raise (flash attention never materialises them).
"""
from __future__ import annotations

import json
import os
import types
from typing import Optional

import torch

from . import checkpoint, generation, gpt_vl_model, lib, vision
from .inference_params import ForwardStep


def _get(cfg, name, default=None):
    return cfg.get(name, default) if isinstance(cfg, dict) else getattr(cfg, name, default)


def configs_from_hf(config):
    """LongVITAConfig (object or the dict of config.json; config_14B.json:1-56) -> (GPTConfig, VisionConfig)."""
    heads = _get(config, "num_attention_heads")
    hidden = _get(config, "hidden_size")
    gcfg = gpt_vl_model.GPTConfig(num_layers=_get(config, "num_hidden_layers"), hidden=hidden, heads=heads,
                                  kv_groups=_get(config, "num_key_value_heads", heads), head_dim=_get(config, "head_dim", None) or hidden // heads,
                                  ffn=_get(config, "intermediate_size"), vocab=_get(config, "vocab_size"),
                                  eps=_get(config, "rms_norm_eps", 1e-6), rope_theta=float(_get(config, "rope_theta", 1e6)))
    v = _get(config, "visual")
    if v is None:
        return gcfg, None
    if _get(v, "qk_normalization", False):
        raise NotImplementedError("InternViT-6B's qk_normalization is not on the Long-VITA path (config_14B.json: false)")
    if _get(v, "norm_type", "layer_norm") != "layer_norm" or _get(v, "hidden_act", "gelu") != "gelu":
        raise NotImplementedError("InternViT-300M is layer_norm + gelu (config_14B.json)")
    vh, vheads = _get(v, "hidden_size"), _get(v, "num_attention_heads")
    vcfg = vision.VisionConfig(num_layers=_get(v, "num_hidden_layers"), hidden=vh, heads=vheads, head_dim=vh // vheads,
                               ffn=_get(v, "intermediate_size"), patch=_get(v, "patch_size"), image=_get(v, "image_size"),
                               ln_eps=_get(v, "layer_norm_eps", 1e-6), proj_ln_eps=1e-5,           # torch.nn.LayerNorm default, resampler_projector.py:17
                               llm_hidden=hidden)
    return gcfg, vcfg


class HipCache:
    """What `past_key_values` is on this path: the decode-loop state of inference_params.py (the KV cache shard + offsets)."""

    def __init__(self, step: ForwardStep):
        self.step = step

    def get_seq_length(self, layer_idx: int = 0) -> int:
        return int(self.step.inference_params.sequence_len_offset)

    def __len__(self):
        return len(self.step.inference_params.key_value_memory_dict)


def _output(logits, cache):
    try:
        from transformers.modeling_outputs import CausalLMOutputWithPast
        return CausalLMOutputWithPast(loss=None, logits=logits, past_key_values=cache, hidden_states=None, attentions=None)
    except ImportError:                                   # the adaptor itself needs nothing of transformers
        return types.SimpleNamespace(loss=None, logits=logits, past_key_values=cache, hidden_states=None, attentions=None)


class LongVITAForCausalLM:
    """HIP-backed stand-in for modeling_long_vita.py:LongVITAForCausalLM (class surface the reference's script uses)."""

    def __init__(self, config, model: gpt_vl_model.GPTVLModel, generation_config=None):
        self.config = config
        self.model = model
        self.device = model.p["embed"].device
        self.dtype = torch.bfloat16
        self.training = False
        eos = _get(config, "eos_token_id")
        self.generation_config = generation_config or types.SimpleNamespace(
            max_new_tokens=20, do_sample=False, use_cache=True, top_k=0, top_p=0.0, temperature=1.0, eos_token_id=eos,
            pad_token_id=_get(config, "pad_token_id"))
        self.cache_headroom = 4096          # rows a forward(use_cache=True) prefill leaves for generated tokens (generate() sizes it exactly)

    # -- construction ---------------------------------------------------------------------------------
    @classmethod
    def from_state_dict(cls, config, state_dict, device="cuda", generation_config=None):
        """state_dict: `LongVITAForCausalLM.state_dict()` names — `model.embed_tokens / layers.N.* / norm`, `lm_head`,
        `model.vision_model.*`, `model.vision_projection.*` (modeling_long_vita.py:67-68)."""
        lib.load(allow_build=False)                      # the HIP path or nothing
        gcfg, vcfg = configs_from_hf(config)
        vis = None
        if vcfg is not None and any(k.startswith("model.vision_model.") for k in state_dict):
            vp = checkpoint.hf_vit_to_params(state_dict, vcfg, prefix="model.vision_model.", projector_prefix="model.vision_projection.")
            vis = vision.MegatronVisionModel.from_oracle_layout(vcfg, vp, device)
        lp = checkpoint.hf_llm_to_params(state_dict, gcfg)
        model = gpt_vl_model.GPTVLModel.from_oracle_layout(gcfg, lp, external_feature_model=vis, device=device)
        return cls(config, model, generation_config)

    @classmethod
    def from_pretrained(cls, model_path, *model_args, torch_dtype=None, device_map=None, attn_implementation=None,
                        trust_remote_code=None, device=None, **kwargs):
        """`AutoModelForCausalLM.from_pretrained(model_path, trust_remote_code=True, device_map=..., torch_dtype=torch.bfloat16,
        attn_implementation="flash_attention_2")` of inference_long_vita.py:811-817: config.json + *.safetensors of a `*_HF` directory."""
        if torch_dtype not in (None, torch.bfloat16, "bfloat16", "auto"):
            raise ValueError("the HIP path computes in bf16 (the reference script passes torch_dtype=torch.bfloat16)")
        config = json.load(open(os.path.join(model_path, "config.json")))
        gen = None
        gpath = os.path.join(model_path, "generation_config.json")
        if os.path.exists(gpath):
            g = json.load(open(gpath))
            gen = types.SimpleNamespace(**{**dict(max_new_tokens=20, do_sample=False, use_cache=True, top_k=0, top_p=0.0, temperature=1.0,
                                                  eos_token_id=config.get("eos_token_id"), pad_token_id=config.get("pad_token_id")), **g})
        dev = device or (device_map if isinstance(device_map, (str, torch.device)) and device_map != "auto" else "cuda")
        return cls.from_state_dict(config, checkpoint.load_hf_safetensors(model_path), device=dev, generation_config=gen)

    def eval(self):
        return self

    def to(self, *a, **k):
        return self

    # -- forward --------------------------------------------------------------------------------------
    def _external_inputs(self, images, image_indices):
        if images is None:
            return None
        if self.model.external_feature_model is None:
            raise ValueError("images given to a checkpoint without vision weights")
        return {"images": images.to(self.device, torch.bfloat16), "indices": image_indices.to(self.device)}

    @torch.no_grad()
    def forward(self, input_ids: Optional[torch.Tensor] = None, attention_mask: Optional[torch.Tensor] = None,
                images: Optional[torch.Tensor] = None, image_indices: Optional[torch.Tensor] = None,
                position_ids: Optional[torch.Tensor] = None, past_key_values=None, inputs_embeds: Optional[torch.Tensor] = None,
                labels=None, use_cache: Optional[bool] = None, output_attentions: Optional[bool] = None,
                output_hidden_states: Optional[bool] = None, return_dict: Optional[bool] = None, cache_position=None,
                num_logits_to_keep: int = 0, **kwargs):
        """modeling_long_vita.py:250-327.  logits [1, n, vocab] bf16 for the last `num_logits_to_keep` rows (0: every row, as there)."""
        if (input_ids is None) == (inputs_embeds is None):
            raise ValueError("You must specify exactly one of input_ids or inputs_embeds")                 # :127-128
        if labels is not None:
            raise NotImplementedError("the loss / backward path is pretrain_long_vita.py's (long_vita_amd.training)")
        if output_attentions or output_hidden_states:
            raise NotImplementedError("flash attention does not materialise attentions / per-layer hidden states")
        ref = input_ids if input_ids is not None else inputs_embeds
        b, s = ref.shape[0], ref.shape[1]
        if b != 1:
            raise ValueError("the Long-VITA path runs batch 1")
        if attention_mask is not None and not bool(attention_mask.to(torch.bool).all()):
            raise ValueError("padded attention masks are not on the Long-VITA path (one unpadded request per call)")
        use_cache = _get(self.config, "use_cache", True) if use_cache is None else use_cache
        cached = isinstance(past_key_values, HipCache) and len(past_key_values) > 0
        if past_key_values is not None and not isinstance(past_key_values, HipCache) and len(past_key_values) > 0:
            raise TypeError("past_key_values must come from this model's own forward(use_cache=True)")
        if input_ids is not None:
            input_ids = input_ids.to(self.device)
        if cached:                                                                       # decode steps: images are not re-encoded (:91)
            step = past_key_values.step
            if inputs_embeds is not None:
                raise NotImplementedError("cached decode steps take input_ids")
            off = step.inference_params.sequence_len_offset
            pos = position_ids.to(self.device) if position_ids is not None else torch.arange(off, off + s, device=self.device)[None]
            step.inference_params.logit_mask = None
            logits = step(input_ids, pos, None)                                         # [1, s, V]
            if num_logits_to_keep:
                logits = logits[:, -num_logits_to_keep:]
            out = _output(logits, past_key_values)
            return out if return_dict is not False else (out.logits, out.past_key_values)
        pos = position_ids.to(self.device) if position_ids is not None else torch.arange(s, device=self.device)[None]
        mask = None
        if num_logits_to_keep:
            mask = torch.zeros(1, s, dtype=torch.bool, device=self.device)
            mask[0, -num_logits_to_keep:] = True
        ext = self._external_inputs(images, image_indices)
        dec_in = None
        if inputs_embeds is not None:                                                    # the caller embedded (and scattered) already
            if ext is not None:
                raise ValueError("inputs_embeds and images together: scatter the features into the embeddings yourself, or pass input_ids")
            dec_in = inputs_embeds.to(self.device, torch.bfloat16).transpose(0, 1).contiguous()        # [b, s, h] -> [s, b, h]
            input_ids = torch.zeros(1, s, dtype=torch.long, device=self.device)
        if use_cache:
            step = ForwardStep(self.model, 1, s + self.cache_headroom)
            ip = step.inference_params
            ip.external_inputs, ip.logit_mask = ext, mask
            if dec_in is not None:
                raise NotImplementedError("inputs_embeds with use_cache=True")
            logits = step(input_ids, pos, None)
            cache = HipCache(step)
        else:
            logits = self.model(input_ids, pos, None, decoder_input=dec_in, external_inputs=ext, logit_mask=mask)
            cache = None
        out = _output(logits, cache)
        return out if return_dict is not False else (out.logits, out.past_key_values)

    __call__ = forward

    # -- generate -------------------------------------------------------------------------------------
    @torch.no_grad()
    def generate(self, inputs: Optional[torch.Tensor] = None, images: Optional[torch.Tensor] = None,
                 image_indices: Optional[torch.Tensor] = None, input_ids: Optional[torch.Tensor] = None, generation_config=None,
                 max_new_tokens: Optional[int] = None, do_sample: Optional[bool] = None, eos_token_id=None, **kwargs):
        """`model.generate(inputs=inputs, images=images, image_indices=image_indices)` (inference_long_vita.py:868): one prefill over
        the prompt (ViT + scatter + decoder), then one cached step per token; stops behind the first end-of-sequence token.  Returns
        [1, prompt + generated] like transformers' greedy search."""
        ids = inputs if inputs is not None else input_ids
        if ids is None or ids.dim() != 2 or ids.shape[0] != 1:
            raise ValueError("generate(inputs=[1, S] token ids) — the Long-VITA path runs batch 1")
        gc = generation_config or self.generation_config
        n_new = max_new_tokens if max_new_tokens is not None else (_get(gc, "max_new_tokens") or 20)
        do_sample = _get(gc, "do_sample", False) if do_sample is None else do_sample
        eos = eos_token_id if eos_token_id is not None else _get(gc, "eos_token_id")
        eos = set() if eos is None else set(eos) if isinstance(eos, (list, tuple, set)) else {int(eos)}
        ids = ids.to(self.device)
        s = ids.shape[1]
        pad = _get(gc, "pad_token_id")
        tokens = torch.full((1, s + n_new), 0 if pad is None else int(pad), dtype=torch.long, device=self.device)
        tokens[:, :s] = ids
        lengths = torch.tensor([s], dtype=torch.long, device=self.device)
        out = tokens[:, :s]
        loop = generation.generate_tokens_probs_and_return_on_first_stage(
            self.model, tokens, lengths, do_sample=bool(do_sample), top_k=int(_get(gc, "top_k", 0) or 0) if do_sample else 0,
            top_p=float(_get(gc, "top_p", 0.0) or 0.0) if do_sample else 0.0, temperature=float(_get(gc, "temperature", 1.0) or 1.0),
            external_inputs=self._external_inputs(images, image_indices), use_kv_cache=bool(_get(gc, "use_cache", True)), logit_mask=True,
            termination_id=None)
        for out, _, _ in loop:
            if int(out[0, -1]) in eos:
                break
        return out.clone()
