"""A stand-in for the pieces of Megatron-LM (core_r0.7.0, the submodule the reference pins at R/.gitmodules:4-6 and does
not vendor) that BUILD and CALL the modules this repo replaces — test infrastructure only.

`install()` puts stub `megatron.*` modules into sys.modules:
  * spec_utils.ModuleSpec / build_module          (how Megatron constructs every layer: `Module(*args, **params, submodules=)`)
  * TransformerLayer / SelfAttention / MLP        the WIRING of a decoder layer, restated from Megatron's published sources:
      the constructor arguments each passes to its submodules (`linear_qkv(hidden, q + 2 kv, config=, init_method=,
      gather_output=False, bias=add_bias_linear or add_qkv_bias, skip_bias_add=False, is_expert=False,
      tp_comm_buffer_name='qkv')`, `core_attention(config=, layer_number=, attn_mask_type=, attention_type=)`,
      `linear_proj(q, hidden, ..., input_is_parallel=True, skip_bias_add=True)`, the MLP's `linear_fc1(hidden, 2 ffn, ...,
      skip_bias_add=True)` / `linear_fc2`), the `[sq, b, ng, (np/ng + 2) hn]` split of the mixed QKV, RoPE through
      `apply_rotary_pos_emb(t, freqs, config=, cu_seqlens=)`, residuals through `bias_dropout_add`;
  * enums.AttnMaskType, identity_op.IdentityOp, a parallel_state with TP = CP = 1, TransformerConfig as a plain dataclass.
No arithmetic lives here except the SwiGLU of Megatron's MLP (`F.silu(gate) * up`, computed through the product's SwiGLUFn so
that the layer stays on the library) — the point is that the product's modules survive Megatron's own construction calls.
"""
from __future__ import annotations

import enum
import sys
import types
from dataclasses import dataclass, field
from typing import Callable, Optional, Union

import torch


class AttnMaskType(enum.Enum):
    padding = 1
    causal = 2
    no_mask = 3


@dataclass
class ModuleSpec:
    module: Union[tuple, type]
    params: dict = field(default_factory=lambda: {})
    submodules: object = None


def build_module(spec_or_module, *args, **kwargs):
    """megatron.core.transformer.spec_utils.build_module."""
    if isinstance(spec_or_module, types.FunctionType):
        return spec_or_module
    if isinstance(spec_or_module, ModuleSpec) and isinstance(spec_or_module.module, types.FunctionType):
        return spec_or_module.module
    if isinstance(spec_or_module, type):
        module = spec_or_module
    elif hasattr(spec_or_module, "module") and isinstance(spec_or_module.module, type):
        module = spec_or_module.module
    else:
        raise TypeError(spec_or_module)
    if hasattr(spec_or_module, "submodules") and spec_or_module.submodules is not None:
        kwargs["submodules"] = spec_or_module.submodules
    return module(*args, **(spec_or_module.params if hasattr(spec_or_module, "params") else {}), **kwargs)


class IdentityOp(torch.nn.Module):
    def __init__(self, *args, **kwargs):
        super().__init__()

    def forward(self, x, *args, **kwargs):
        return x


class IdentityFuncOp(IdentityOp):
    """megatron.core.transformer.identity_op.IdentityFuncOp: forward returns the identity FUNCTION (the default bias-dropout-add)."""

    def forward(self, *args, **kwargs):
        return super().forward


@dataclass
class TransformerConfig:
    num_layers: int = 1
    hidden_size: int = 1024
    num_attention_heads: int = 8
    num_query_groups: Optional[int] = 2
    kv_channels: int = 128
    ffn_hidden_size: int = 2816
    normalization: str = "RMSNorm"
    layernorm_epsilon: float = 1e-6
    add_bias_linear: bool = False
    add_qkv_bias: bool = True
    gated_linear_unit: bool = True
    activation_func: Callable = torch.nn.functional.silu
    attention_dropout: float = 0.0
    hidden_dropout: float = 0.0
    bias_dropout_fusion: bool = False
    params_dtype: torch.dtype = torch.bfloat16
    use_cpu_initialization: bool = False
    perform_initialization: bool = True
    sequence_parallel: bool = False
    gradient_accumulation_fusion: bool = False
    tensor_model_parallel_size: int = 1
    expert_model_parallel_size: int = 1
    context_parallel_size: int = 1
    rotary_interleaved: bool = False
    apply_rope_fusion: bool = False
    init_method: Optional[Callable] = None
    output_layer_init_method: Optional[Callable] = None
    init_method_std: float = 0.02

    def __post_init__(self):
        if self.init_method is None:
            self.init_method = lambda w: torch.nn.init.normal_(w, mean=0.0, std=self.init_method_std)
        if self.output_layer_init_method is None:
            self.output_layer_init_method = self.init_method


@dataclass
class SelfAttentionSubmodules:
    linear_qkv: object = None
    core_attention: object = None
    linear_proj: object = None
    q_layernorm: object = None
    k_layernorm: object = None


@dataclass
class MLPSubmodules:
    linear_fc1: object = None
    linear_fc2: object = None


@dataclass
class TransformerLayerSubmodules:
    input_layernorm: object = IdentityOp
    self_attention: object = IdentityOp
    self_attn_bda: object = IdentityFuncOp
    pre_cross_attn_layernorm: object = IdentityOp
    cross_attention: object = IdentityOp
    cross_attn_bda: object = None
    pre_mlp_layernorm: object = IdentityOp
    mlp: object = IdentityOp
    mlp_bda: object = IdentityFuncOp
    sharded_state_dict_keys_map: dict = field(default_factory=dict)


class SelfAttention(torch.nn.Module):
    """megatron.core.transformer.attention.SelfAttention (+ Attention.__init__ / forward), TP = 1, no inference cache."""

    def __init__(self, config, submodules: SelfAttentionSubmodules, layer_number: int, attn_mask_type=AttnMaskType.padding):
        super().__init__()
        self.config, self.layer_number, self.attn_mask_type, self.attention_type = config, layer_number, attn_mask_type, "self"
        self.query_projection_size = config.kv_channels * config.num_attention_heads
        self.kv_projection_size = config.kv_channels * config.num_query_groups
        self.hidden_size_per_attention_head = config.kv_channels
        from megatron.core import parallel_state
        world = parallel_state.get_tensor_model_parallel_world_size()
        self.num_attention_heads_per_partition = config.num_attention_heads // world
        self.num_query_groups_per_partition = config.num_query_groups // world
        self.core_attention = build_module(submodules.core_attention, config=config, layer_number=layer_number,
                                           attn_mask_type=attn_mask_type, attention_type="self")
        self.linear_proj = build_module(submodules.linear_proj, self.query_projection_size, config.hidden_size, config=config,
                                        init_method=config.output_layer_init_method, bias=config.add_bias_linear,
                                        input_is_parallel=True, skip_bias_add=True, is_expert=False, tp_comm_buffer_name="proj")
        self.linear_qkv = build_module(submodules.linear_qkv, config.hidden_size,
                                       self.query_projection_size + 2 * self.kv_projection_size, config=config,
                                       init_method=config.init_method, gather_output=False,
                                       bias=config.add_bias_linear or config.add_qkv_bias, skip_bias_add=False, is_expert=False,
                                       tp_comm_buffer_name="qkv")
        self.q_layernorm = build_module(submodules.q_layernorm) if submodules.q_layernorm is IdentityOp else None
        self.k_layernorm = None

    def get_query_key_value_tensors(self, hidden_states):
        mixed_qkv, _ = self.linear_qkv(hidden_states)                                    # [sq, b, ng (np/ng + 2) hn]
        ng, hn = self.num_query_groups_per_partition, self.hidden_size_per_attention_head
        qpg = self.num_attention_heads_per_partition // ng
        mixed_qkv = mixed_qkv.view(*mixed_qkv.shape[:-1], ng, (qpg + 2) * hn)
        query, key, value = torch.split(mixed_qkv, [qpg * hn, hn, hn], dim=3)
        query = query.reshape(query.size(0), query.size(1), -1, hn)                      # [sq, b, np, hn]
        return query, key, value

    def forward(self, hidden_states, attention_mask=None, inference_params=None, rotary_pos_emb=None, packed_seq_params=None):
        from megatron.core.models.common.embeddings.rotary_pos_embedding import apply_rotary_pos_emb
        query, key, value = self.get_query_key_value_tensors(hidden_states)
        if rotary_pos_emb is not None:
            if not isinstance(rotary_pos_emb, tuple):
                rotary_pos_emb = (rotary_pos_emb,) * 2
            q_pos_emb, k_pos_emb = rotary_pos_emb
            query = apply_rotary_pos_emb(query.contiguous(), q_pos_emb, config=self.config, cu_seqlens=None)
            key = apply_rotary_pos_emb(key.contiguous(), k_pos_emb, config=self.config, cu_seqlens=None)
        core_attn_out = self.core_attention(query, key, value.contiguous(), attention_mask, attn_mask_type=self.attn_mask_type,
                                            packed_seq_params=packed_seq_params)
        return self.linear_proj(core_attn_out)                                           # (output, bias)


class MLP(torch.nn.Module):
    """megatron.core.transformer.mlp.MLP."""

    def __init__(self, config, submodules: MLPSubmodules, is_expert: bool = False, input_size: int = None):
        super().__init__()
        self.config = config
        self.input_size = input_size if input_size is not None else config.hidden_size
        ffn = config.ffn_hidden_size * (2 if config.gated_linear_unit else 1)
        self.linear_fc1 = build_module(submodules.linear_fc1, self.input_size, ffn, config=config, init_method=config.init_method,
                                       gather_output=False, bias=config.add_bias_linear, skip_bias_add=True, is_expert=is_expert,
                                       tp_comm_buffer_name="fc1")
        self.activation_func = config.activation_func
        self.linear_fc2 = build_module(submodules.linear_fc2, config.ffn_hidden_size, config.hidden_size, config=config,
                                       init_method=config.output_layer_init_method, bias=config.add_bias_linear,
                                       input_is_parallel=True, skip_bias_add=True, is_expert=is_expert, tp_comm_buffer_name="fc2")

    def forward(self, hidden_states):
        intermediate_parallel, bias_parallel = self.linear_fc1(hidden_states)
        assert bias_parallel is None and self.config.gated_linear_unit and self.activation_func is torch.nn.functional.silu
        # Megatron: glu(x) = silu(chunk0) * chunk1 — one library kernel (vita_swiglu_fwd / _bwd) instead of three torch ops
        from long_vita_amd.autograd_fns import SwiGLUFn
        intermediate_parallel = SwiGLUFn.apply(intermediate_parallel)
        return self.linear_fc2(intermediate_parallel)                                    # (output, bias)


class TransformerLayer(torch.nn.Module):
    """megatron.core.transformer.transformer_layer.TransformerLayer (self-attention decoder layer, no cross attention)."""

    def __init__(self, config, submodules: TransformerLayerSubmodules, layer_number: int = 1, hidden_dropout: float = None):
        super().__init__()
        self.config, self.layer_number = config, layer_number
        self.hidden_dropout = config.hidden_dropout if hidden_dropout is None else hidden_dropout
        self.input_layernorm = build_module(submodules.input_layernorm, config=config, hidden_size=config.hidden_size,
                                            eps=config.layernorm_epsilon)
        self.self_attention = build_module(submodules.self_attention, config=config, layer_number=layer_number)
        self.self_attn_bda = build_module(submodules.self_attn_bda)
        self.pre_mlp_layernorm = build_module(submodules.pre_mlp_layernorm, config=config, hidden_size=config.hidden_size,
                                              eps=config.layernorm_epsilon)
        self.mlp = build_module(submodules.mlp, config=config)
        self.mlp_bda = build_module(submodules.mlp_bda)

    def forward(self, hidden_states, attention_mask=None, context=None, context_mask=None, rotary_pos_emb=None,
                inference_params=None, packed_seq_params=None):
        residual = hidden_states
        x = self.input_layernorm(hidden_states)
        attention_output_with_bias = self.self_attention(x, attention_mask=attention_mask, inference_params=inference_params,
                                                         rotary_pos_emb=rotary_pos_emb, packed_seq_params=packed_seq_params)
        hidden_states = self.self_attn_bda(self.training, self.config.bias_dropout_fusion)(attention_output_with_bias, residual,
                                                                                           self.hidden_dropout)
        residual = hidden_states
        x = self.pre_mlp_layernorm(hidden_states)
        mlp_output_with_bias = self.mlp(x)
        hidden_states = self.mlp_bda(self.training, self.config.bias_dropout_fusion)(mlp_output_with_bias, residual,
                                                                                     self.hidden_dropout)
        return hidden_states, context


_STUBS = {
    "megatron.core.transformer.spec_utils": dict(ModuleSpec=ModuleSpec, build_module=build_module),
    "megatron.core.transformer.enums": dict(AttnMaskType=AttnMaskType),
    "megatron.core.transformer.identity_op": dict(IdentityOp=IdentityOp, IdentityFuncOp=IdentityFuncOp),
    "megatron.core.transformer.attention": dict(SelfAttention=SelfAttention, SelfAttentionSubmodules=SelfAttentionSubmodules),
    "megatron.core.transformer.mlp": dict(MLP=MLP, MLPSubmodules=MLPSubmodules),
    "megatron.core.transformer.transformer_layer": dict(TransformerLayer=TransformerLayer,
                                                        TransformerLayerSubmodules=TransformerLayerSubmodules),
    "megatron.core.transformer.transformer_config": dict(TransformerConfig=TransformerConfig),
}


def install():
    """Create the stub tree in sys.modules (parents included); returns the list of names to pop again."""
    names = set()
    for full in list(_STUBS) + ["megatron.core.models.gpt.gpt_layer_specs", "megatron.core.models.common.embeddings.rotary_pos_embedding",
                                "megatron.core.models.common.embeddings.language_model_embedding", "megatron.core.tensor_parallel.layers",
                                "megatron.core.transformer.dot_product_attention", "megatron.core.parallel_state",
                                "long_vita_megatron.core.models.vision.vit_layer_specs"]:
        parts = full.split(".")
        for i in range(1, len(parts) + 1):
            names.add(".".join(parts[:i]))
    for n in sorted(names, key=len):
        if n not in sys.modules:
            sys.modules[n] = types.ModuleType(n)
        if "." in n:
            parent, leaf = n.rsplit(".", 1)
            setattr(sys.modules[parent], leaf, sys.modules[n])
    for mod, attrs in _STUBS.items():
        for k, v in attrs.items():
            setattr(sys.modules[mod], k, v)

    class _UpstreamDPA:                                   # the class whose .forward the adaptor wraps (:21-22)
        def forward(self, *a, **k):
            raise AssertionError("Megatron's unfused attention must not run")
    sys.modules["megatron.core.transformer.dot_product_attention"].DotProductAttention = _UpstreamDPA
    ps = sys.modules["megatron.core.parallel_state"]
    ps.model_parallel_is_initialized = lambda: False     # long_vita_amd.parallel_state's own (thread-local, test-set) state answers
    from long_vita_amd import parallel_state as own
    ps.get_tensor_model_parallel_world_size = own.get_tensor_model_parallel_world_size
    ps.get_tensor_model_parallel_rank = own.get_tensor_model_parallel_rank
    return sorted(names)


def uninstall(names):
    for n in names:
        sys.modules.pop(n, None)
