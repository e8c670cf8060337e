"""Layer-spec builders — the drop-in for M/core/models/gpt/gpt_layer_specs.py:26-104, registered by megatron_adaptor on
`megatron.core.models.gpt.gpt_layer_specs.get_gpt_layer_local_spec` / `..._with_transformer_engine_spec`
(M/megatron_adaptor.py:81-88).

The reference returns `ModuleSpec(module=TransformerLayer, submodules=...)` whose leaves are TransformerEngine / Megatron
modules; here the leaves are the HIP-backed modules of this package, so that under a real Megatron the decoder's norms, the
four linears and the core attention — every kernel of the layer — run through libvita_hip.so:

  both specs           mlp = layers.GatedMLP (Megatron MLP's constructor; SwiGLU as the fc1 GEMM epilogue / vita_swiglu_fwd,bwd)
  TE spec   (:26-55)   linear_qkv = LayerNormColumnParallelLinear   (vita_rmsnorm_fwd -> vita_gemm_bf16, + bias)
                       core_attention = HipDotProductAttention      (vita_flash_attn_fwd / _bwd)
                       linear_proj = RowParallelLinear, mlp.linear_fc1 = LayerNormColumnParallelLinear, mlp.linear_fc2 = RowParallelLinear
  local spec (:59-90)  input_layernorm / pre_mlp_layernorm = Norm (PTNorm), linear_qkv / linear_fc1 = ColumnParallelLinear, ...
                       with the reference's sharded_state_dict_keys_map

TransformerLayer, SelfAttention, MLP, ModuleSpec and the submodule dataclasses are Megatron's own (they only wire the
leaves together); they are imported when the builder is called, so this module imports without Megatron.
"""
from __future__ import annotations

from .dot_product_attention import HipDotProductAttention
from .layers import (ColumnParallelLinear, GatedMLP, LayerNormColumnParallelLinear, Norm, RowParallelLinear, get_bias_dropout_add)


def _megatron():
    from megatron.core.transformer.attention import SelfAttention, SelfAttentionSubmodules
    from megatron.core.transformer.enums import AttnMaskType
    from megatron.core.transformer.identity_op import IdentityOp
    from megatron.core.transformer.mlp import MLP, MLPSubmodules
    from megatron.core.transformer.spec_utils import ModuleSpec
    from megatron.core.transformer.transformer_layer import TransformerLayer, TransformerLayerSubmodules
    return dict(SelfAttention=SelfAttention, SelfAttentionSubmodules=SelfAttentionSubmodules, AttnMaskType=AttnMaskType,
                IdentityOp=IdentityOp, MLP=MLP, MLPSubmodules=MLPSubmodules, ModuleSpec=ModuleSpec,
                TransformerLayer=TransformerLayer, TransformerLayerSubmodules=TransformerLayerSubmodules)


def _get_mlp_module_spec(use_te: bool = True, num_experts: int = None, moe_grouped_gemm: bool = False):
    """:93-104 — dense MLP only (Long-VITA has no experts)."""
    if num_experts is not None:
        raise NotImplementedError("mixture-of-experts layers are not on the Long-VITA path")
    m = _megatron()
    # module = GatedMLP (Megatron MLP's constructor and return value): the gated activation stays on the library — the fc1 GEMM's
    # epilogue without autograd, vita_swiglu_fwd / _bwd with it — instead of Megatron's three torch ops on the 2 ffn-wide product
    return m["ModuleSpec"](module=GatedMLP, submodules=m["MLPSubmodules"](
        linear_fc1=LayerNormColumnParallelLinear if use_te else ColumnParallelLinear, linear_fc2=RowParallelLinear))


def get_gpt_layer_with_transformer_engine_spec(num_experts: int = None, moe_grouped_gemm: bool = False, qk_layernorm: bool = False):
    """:26-55 with HIP-backed leaves.  Norms are folded into the following linear (`layer_norm_weight` beside `weight`)."""
    m = _megatron()
    mlp = _get_mlp_module_spec(use_te=True, num_experts=num_experts, moe_grouped_gemm=moe_grouped_gemm)
    return m["ModuleSpec"](
        module=m["TransformerLayer"],
        submodules=m["TransformerLayerSubmodules"](
            self_attention=m["ModuleSpec"](
                module=m["SelfAttention"],
                params={"attn_mask_type": m["AttnMaskType"].causal},
                submodules=m["SelfAttentionSubmodules"](
                    linear_qkv=LayerNormColumnParallelLinear,
                    core_attention=HipDotProductAttention,
                    linear_proj=RowParallelLinear,
                    q_layernorm=Norm if qk_layernorm else m["IdentityOp"],
                    k_layernorm=Norm if qk_layernorm else m["IdentityOp"],
                ),
            ),
            self_attn_bda=get_bias_dropout_add,
            pre_mlp_layernorm=m["IdentityOp"],
            mlp=mlp,
            mlp_bda=get_bias_dropout_add,
        ),
    )


def get_gpt_layer_local_spec(num_experts: int = None, moe_grouped_gemm: bool = False, qk_layernorm: bool = False):
    """:59-90 with HIP-backed leaves (separate norm modules, the reference's sharded_state_dict_keys_map)."""
    m = _megatron()
    mlp = _get_mlp_module_spec(use_te=False, num_experts=num_experts, moe_grouped_gemm=moe_grouped_gemm)
    return m["ModuleSpec"](
        module=m["TransformerLayer"],
        submodules=m["TransformerLayerSubmodules"](
            input_layernorm=Norm,
            self_attention=m["ModuleSpec"](
                module=m["SelfAttention"],
                params={"attn_mask_type": m["AttnMaskType"].causal},
                submodules=m["SelfAttentionSubmodules"](
                    linear_qkv=ColumnParallelLinear,
                    core_attention=HipDotProductAttention,
                    linear_proj=RowParallelLinear,
                    q_layernorm=Norm if qk_layernorm else m["IdentityOp"],
                    k_layernorm=Norm if qk_layernorm else m["IdentityOp"],
                ),
            ),
            self_attn_bda=get_bias_dropout_add,
            pre_mlp_layernorm=Norm,
            mlp=mlp,
            mlp_bda=get_bias_dropout_add,
            sharded_state_dict_keys_map={
                'input_layernorm.': 'self_attention.linear_qkv.layer_norm_',
                'pre_mlp_layernorm.': 'mlp.linear_fc1.layer_norm_',
            },
        ),
    )
