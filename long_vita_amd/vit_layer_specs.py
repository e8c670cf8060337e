"""ViT layer-spec builders — the drop-in for M/core/models/vision/vit_layer_specs.py:30-101, registered by megatron_adaptor on
`long_vita_megatron.core.models.vision.vit_layer_specs.get_vit_layer_local_spec_for_intern` (the builder
MegatronVisionModel.__init__ calls: M/pretrain_long_vita.py:337-356 hard-codes `use_te = False`), `..._with_transformer_engine_spec_for_intern`
and `get_vit_layer_local_spec_for_siglip` (:366-370).

The reference returns `ModuleSpec(module=InternViTTransformerLayer | SigLIPViTTransformerLayer, submodules=...)` whose leaves are
Megatron / TransformerEngine modules and whose layer classes do the bias add, the LayerScale multiply and the residual add as
three torch ops.  Here every leaf is a HIP-backed module of this package, so under a real Megatron the ViT's norms, its four
linears, the non-causal core attention, the GELU and the LayerScale residual all run through libvita_hip.so — forward and backward:

  input_layernorm / pre_mlp_layernorm   layers.Norm (LayerNorm: vita_layernorm_fwd / vita_layernorm_bwd)
  linear_qkv                            layers.ColumnParallelLinear (+ bias), TE spec: layers.LayerNormColumnParallelLinear
  core_attention                        dot_product_attention.HipDotProductAttention (AttnMaskType.no_mask -> d = 64 non-causal kernel;
                                        autograd_fns.FlashAttnNonCausalFn)
  linear_proj, linear_fc2               layers.RowParallelLinear (skip_bias_add: the bias goes into the residual kernel)
  mlp                                   layers.ViTMLP (Megatron MLP's constructor; fc1 + bias + GELU as ONE GEMM epilogue without autograd,
                                        GEMM(+bias) -> vita_gelu_fwd with it)
  layer                                 `residual + (out + bias) * ls` as one kernel (vita_bias_scale_res_fwd / _bwd), parameters `ls1`, `ls2`
                                        under the reference's names (intern_vit_model.py:43-44; checkpoints of L/ckpt_converter_intern_vit.py)

TransformerLayer, SelfAttention, ModuleSpec and the submodule dataclasses are Megatron's own; they are imported when a builder is
called, so this module imports without Megatron.  The layer classes subclass Megatron's TransformerLayer (its __init__ builds the
submodules from the spec), hence they are created on first use."""
from __future__ import annotations

import torch
from torch.nn import Parameter

from . import autograd_fns as F_
from .dot_product_attention import HipDotProductAttention
from .gpt_layer_specs import _megatron
from .layers import ColumnParallelLinear, LayerNormColumnParallelLinear, Norm, RowParallelLinear, ViTMLP, _alloc

_CLASSES = {}


def _layer_classes():
    base = _megatron()["TransformerLayer"]
    if base in _CLASSES:
        return _CLASSES[base]

    class _HipViTLayer(base):
        """hidden [s, b, h] -> LN -> self-attention -> residual kernel -> LN -> MLP -> residual kernel
        (InternViTTransformerLayer.forward, intern_vit_model.py:46-89 / SigLIPViTTransformerLayer.forward, siglip_vit_model.py:29-86)."""
        layerscale = False

        def __init__(self, config, submodules, layer_number: int = 1, hidden_dropout: float = None):
            super().__init__(config=config, submodules=submodules, layer_number=layer_number, hidden_dropout=hidden_dropout)
            if self.layerscale:
                self.ls1 = Parameter(torch.full_like(_alloc(config, config.hidden_size), 0.01))      # intern_vit_model.py:43-44
                self.ls2 = Parameter(torch.full_like(_alloc(config, config.hidden_size), 0.01))

        def forward(self, hidden_states, attention_mask=None, context=None, context_mask=None, rotary_pos_emb=None,
                    inference_params=None, packed_seq_params=None):
            residual = hidden_states
            x = self.input_layernorm(hidden_states)
            attention_output, attention_bias = self.self_attention(x, attention_mask=attention_mask, inference_params=inference_params,
                                                                   rotary_pos_emb=rotary_pos_emb, packed_seq_params=packed_seq_params)
            hidden_states = F_.BiasScaleResidualFn.apply(attention_output, attention_bias, self.ls1 if self.layerscale else None, residual)
            residual = hidden_states
            x = self.pre_mlp_layernorm(hidden_states)
            mlp_output, mlp_bias = self.mlp(x)
            hidden_states = F_.BiasScaleResidualFn.apply(mlp_output, mlp_bias, self.ls2 if self.layerscale else None, residual)
            return hidden_states, context

    class InternViTTransformerLayer(_HipViTLayer):
        layerscale = True

    class SigLIPViTTransformerLayer(_HipViTLayer):
        layerscale = False

    _CLASSES[base] = (InternViTTransformerLayer, SigLIPViTTransformerLayer)
    return _CLASSES[base]


def _vit_spec(layer_cls, fused_norm: bool, unfused_bias: bool = False):
    m = _megatron()
    attn = m["ModuleSpec"](module=m["SelfAttention"], params={"attn_mask_type": m["AttnMaskType"].no_mask},
                           submodules=m["SelfAttentionSubmodules"](
                               linear_qkv=LayerNormColumnParallelLinear if fused_norm else ColumnParallelLinear,
                               core_attention=HipDotProductAttention, linear_proj=RowParallelLinear))
    mlp = m["ModuleSpec"](module=ViTMLP, params={"unfused_bias": unfused_bias}, submodules=m["MLPSubmodules"](
        linear_fc1=LayerNormColumnParallelLinear if fused_norm else ColumnParallelLinear, linear_fc2=RowParallelLinear))
    norm = m["IdentityOp"] if fused_norm else Norm
    return m["ModuleSpec"](module=layer_cls, submodules=m["TransformerLayerSubmodules"](
        self_attention=attn, mlp=mlp, pre_mlp_layernorm=norm, input_layernorm=norm))


def get_vit_layer_local_spec_for_intern(use_te=True):
    """vit_layer_specs.py:79-101 — the spec every reference script builds the InternViT-300M from."""
    return _vit_spec(_layer_classes()[0], fused_norm=False, unfused_bias=True)     # Megatron MLP, bias_activation_fusion off (:213)


def get_vit_layer_with_transformer_engine_spec_for_intern(use_te=True):
    """vit_layer_specs.py:55-77 — norms folded into the following linear (`layer_norm_weight` / `layer_norm_bias` beside `weight`)."""
    return _vit_spec(_layer_classes()[0], fused_norm=True)


def get_vit_layer_local_spec_for_siglip(use_te=True):
    """vit_layer_specs.py:30-53 — SigLIP-400M (M/pretrain_long_vita.py:268-307: hidden 1152, 16 heads x 72, FFN 4304, tanh GELU, no
    LayerScale, no class token).  The sizes the MFMA kernels do not tile are padded where they are used: head size 72 -> 96 with zero
    columns inside HipDotProductAttention (scores and outputs unchanged), the 4304-deep contractions of fc2 and of fc1's dgrad -> 4352
    with zero columns inside ops.gemm; parameters keep Megatron's shapes.  Biases of proj / fc1 / fc2 meet the bf16-rounded product in
    an op of their own (skip_bias_add, no fusion: `unfused_bias`), as the reference's modules do."""
    return _vit_spec(_layer_classes()[1], fused_norm=False, unfused_bias=True)
