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
  * r04: TransformerBlock (the layer stack + `final_layernorm` built from the name `TENorm` of its own module namespace, as
    M/core/transformer/transformer_block.py:16-18,201 imports and calls it; `--recompute-granularity full --recompute-method block
    --recompute-num-layers N` through `tensor_parallel.checkpoint`), tensor_parallel.checkpoint itself (CheckpointFunction of
    megatron/core/tensor_parallel/random.py restated: forward under no_grad keeping the inputs and the RNG states, backward = the
    function re-run with autograd on — the way Megatron RE-ENTERS this package's autograd Functions), `GPTVLModel`: the reference's
    `__init__` (gpt_vl_model.py:73-180) and the training branch of its `forward` (:233-416) restated over the names it imports, with
    `forward_step` / `loss_func` of the entry script (pretrain_long_vita.py:778-869) and Megatron's RotaryEmbedding (angle table), and `MegatronVisionModel`: the class
    of the reference's ENTRY SCRIPT (M/pretrain_long_vita.py:310-520) restated as plain torch — constructor wiring, forward_once /
    forward_chunk / forward, the torch pixel_shuffle and torch.nn.LayerNorm the drop-in has to displace.
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
    recompute_granularity: Optional[str] = None
    recompute_method: Optional[str] = None
    recompute_num_layers: Optional[int] = None

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


class _TENormPlaceholder:
    """megatron.core.transformer.custom_layers.transformer_engine.TENorm: needs TransformerEngine; the adaptor replaces it."""

    def __new__(cls, config, hidden_size, eps=1e-5):
        raise AssertionError("TransformerEngine's TENorm must not be constructed: the adaptor registers layers.Norm on this name")


class CheckpointFunction(torch.autograd.Function):
    """megatron.core.tensor_parallel.random.CheckpointFunction (distribute_saved_activations = False)."""

    @staticmethod
    def forward(ctx, run_function, distribute_saved_activations, *args):
        ctx.run_function = run_function
        ctx.fwd_cpu_rng_state = torch.get_rng_state()
        ctx.fwd_cuda_rng_state = torch.cuda.get_rng_state() if torch.cuda.is_available() else None
        with torch.no_grad():
            outputs = run_function(*args)
        ctx.save_for_backward(*args)
        return outputs

    @staticmethod
    def backward(ctx, *args):
        if not torch.autograd._is_checkpoint_valid():
            raise RuntimeError("Checkpointing is not compatible with .grad(), please use .backward() if possible")
        inputs = ctx.saved_tensors
        bwd_cpu_rng_state = torch.get_rng_state()
        torch.set_rng_state(ctx.fwd_cpu_rng_state)
        if ctx.fwd_cuda_rng_state is not None:
            bwd_cuda_rng_state = torch.cuda.get_rng_state()
            torch.cuda.set_rng_state(ctx.fwd_cuda_rng_state)
        detached_inputs = tuple(x.detach().requires_grad_(x.requires_grad) if isinstance(x, torch.Tensor) else x for x in inputs)
        with torch.enable_grad():
            outputs = ctx.run_function(*detached_inputs)
        torch.set_rng_state(bwd_cpu_rng_state)
        if ctx.fwd_cuda_rng_state is not None:
            torch.cuda.set_rng_state(bwd_cuda_rng_state)
        if isinstance(outputs, torch.Tensor):
            outputs = (outputs,)
        outputs, args = zip(*filter(lambda x: torch.is_tensor(x[0]), zip(outputs, args)))
        torch.autograd.backward(outputs, args)
        grads = tuple(inp.grad if isinstance(inp, torch.Tensor) else inp for inp in detached_inputs)
        return (None, None) + grads


def checkpoint(function, distribute_saved_activations, *args):
    """megatron.core.tensor_parallel.checkpoint."""
    return CheckpointFunction.apply(function, distribute_saved_activations, *args)


class TransformerBlock(torch.nn.Module):
    """megatron.core.transformer.transformer_block.TransformerBlock: `num_layers` layers from one spec, the final norm through the
    module-level name TENorm, activation recompute per Megatron's `_checkpointed_forward` (method "block": the first
    recompute_num_layers layers are checkpointed one by one; "uniform": chunks of recompute_num_layers layers)."""

    def __init__(self, config, spec, post_layer_norm: bool = True, pre_process: bool = True, post_process: bool = True):
        super().__init__()
        self.config, self.pre_process, self.post_process, self.input_tensor = config, pre_process, post_process, None
        self.layers = torch.nn.ModuleList([build_module(spec, config=config, layer_number=i + 1) for i in range(config.num_layers)])
        self.num_layers_per_pipeline_rank = len(self.layers)
        if post_process and post_layer_norm:
            norm = sys.modules["megatron.core.transformer.transformer_block"].TENorm          # transformer_block.py:16-18,201
            self.final_layernorm = norm(config=config, hidden_size=config.hidden_size, eps=config.layernorm_epsilon)

    def set_input_tensor(self, input_tensor):
        self.input_tensor = input_tensor

    def _checkpointed_forward(self, hidden_states, attention_mask, rotary_pos_emb, packed_seq_params):
        tp = sys.modules["megatron.core.tensor_parallel"]

        def custom(start, end):
            def custom_forward(hidden_states, attention_mask, rotary_pos_emb):
                for index in range(start, end):
                    hidden_states, _ = self.layers[index](hidden_states, attention_mask=attention_mask, rotary_pos_emb=rotary_pos_emb,
                                                          packed_seq_params=packed_seq_params)
                return hidden_states
            return custom_forward

        n, k = len(self.layers), self.config.recompute_num_layers
        if self.config.recompute_method == "uniform":
            for l in range(0, n, k):
                hidden_states = tp.checkpoint(custom(l, min(l + k, n)), False, hidden_states, attention_mask, rotary_pos_emb)
        elif self.config.recompute_method == "block":
            for l in range(n):
                if l < k:
                    hidden_states = tp.checkpoint(custom(l, l + 1), False, hidden_states, attention_mask, rotary_pos_emb)
                else:
                    hidden_states = custom(l, l + 1)(hidden_states, attention_mask, rotary_pos_emb)
        else:
            raise ValueError("Invalid activation recompute method.")
        return hidden_states

    def forward(self, hidden_states, attention_mask=None, context=None, context_mask=None, rotary_pos_emb=None, inference_params=None,
                packed_seq_params=None):
        if not self.pre_process:
            hidden_states = self.input_tensor
        if getattr(self.config, "recompute_granularity", None) == "full" and self.training:
            hidden_states = self._checkpointed_forward(hidden_states, attention_mask, rotary_pos_emb, packed_seq_params)
        else:
            for layer in self.layers:
                hidden_states, context = layer(hidden_states, attention_mask=attention_mask, rotary_pos_emb=rotary_pos_emb,
                                               inference_params=inference_params, packed_seq_params=packed_seq_params)
        if self.post_process and hasattr(self, "final_layernorm"):
            hidden_states = self.final_layernorm(hidden_states)
        return hidden_states


class RotaryEmbedding(torch.nn.Module):
    """megatron.core.models.common.embeddings.rotary_pos_embedding.RotaryEmbedding (the reference leaves it unpatched,
    M/megatron_adaptor.py:102-103): the fp32 ANGLE tensor [s, 1, 1, dim] every layer's apply_rotary_pos_emb receives.  CP = 1 here."""

    def __init__(self, kv_channels, rotary_percent=1.0, rotary_interleaved=False, seq_len_interpolation_factor=None, rotary_base=10000):
        super().__init__()
        dim = kv_channels if rotary_percent >= 1.0 else int(kv_channels * rotary_percent)
        self.inv_freq = 1.0 / (rotary_base ** (torch.arange(0, dim, 2, dtype=torch.float32) / dim))

    def forward(self, max_seq_len, offset=0):
        from long_vita_amd import parallel_state as own          # (megatron.core.parallel_state in the original)
        seq = torch.arange(max_seq_len, dtype=torch.float32) + offset
        freqs = torch.outer(seq, self.inv_freq)
        emb = torch.cat((freqs, freqs), dim=-1)[:, None, None, :]
        cp_size = own.get_context_parallel_world_size()
        if cp_size > 1:                                           # get_pos_emb_on_this_cp_rank: the rank's two zig-zag chunks
            cp_idx = torch.tensor([own.get_context_parallel_rank(), 2 * cp_size - own.get_context_parallel_rank() - 1])
            emb = emb.view(2 * cp_size, -1, *emb.shape[1:]).index_select(0, cp_idx).view(-1, *emb.shape[1:])
        return emb

    def get_rotary_seq_len(self, inference_params, transformer, transformer_input, transformer_config):
        n = transformer_input.size(0) if transformer_input is not None else transformer.input_tensor.size(0)
        return n * transformer_config.context_parallel_size


ARGS = types.SimpleNamespace(output_multiplier_scale=None, output_logit_softcapping=None, is_instruction_dataset=True,
                             context_parallel_size=1)           # what get_args() returns to GPTVLModel.forward / loss_func


class GPTVLModel(torch.nn.Module):
    """long_vita_megatron.core.models.multimodal.gpt_vl_model.GPTVLModel restated: `__init__` (:73-180 — external_feature_model from
    its provider (:110-113), LanguageModelEmbedding, RotaryEmbedding, TransformerBlock, the ColumnParallelLinear output layer, `unused`)
    and `forward` (:233-416: training and inference branches, incl. the `inference_params` overrides :258-262, :278-283), the three
    freeze methods (:182-216).  Pinned bit for bit on the reference's own class over plain-torch leaves (tests/golden/gptvl_forward.pt,
    tests/test_oracle_golden.py).  Every class it instantiates is looked up under the dotted name the reference
    imports it from, i.e. it is whatever the adaptors left there.  transformer_layer_spec = None keeps the r04 shell (vision tests):
    only the external feature model is built."""

    def __init__(self, config, transformer_layer_spec=None, vocab_size=0, max_sequence_length=0, pre_process=True, post_process=True,
                 fp16_lm_cross_entropy=False, parallel_output=True, share_embeddings_and_output_weights=False,
                 position_embedding_type="learned_absolute", rotary_percent=1.0, rotary_base=10000, seq_len_interpolation_factor=None,
                 external_feature_model_provider=None, external_args=(), allow_missing_keys=(), **kwargs):
        super().__init__()
        self.config, self.pre_process, self.post_process = config, pre_process, post_process
        self.vocab_size, self.max_sequence_length = vocab_size, max_sequence_length
        self.position_embedding_type, self.share_embeddings_and_output_weights = position_embedding_type, share_embeddings_and_output_weights
        if pre_process:
            self.external_feature_model = external_feature_model_provider(config, *external_args)                     # :110-113
        if transformer_layer_spec is None:
            return
        if pre_process:
            emb = sys.modules["megatron.core.models.common.embeddings.language_model_embedding"].LanguageModelEmbedding
            self.embedding = emb(config=config, vocab_size=vocab_size, max_sequence_length=max_sequence_length,
                                 position_embedding_type=position_embedding_type)                                      # :115-120
        if position_embedding_type == "rope":
            self.rotary_pos_emb = RotaryEmbedding(kv_channels=config.kv_channels, rotary_percent=rotary_percent,
                                                  rotary_interleaved=config.rotary_interleaved,
                                                  seq_len_interpolation_factor=seq_len_interpolation_factor, rotary_base=rotary_base)
        block = sys.modules["megatron.core.transformer.transformer_block"].TransformerBlock
        self.decoder = block(config=config, spec=transformer_layer_spec, pre_process=pre_process, post_process=post_process)   # :132-137
        if post_process:
            # `tensor_parallel.ColumnParallelLinear`: the package re-exports megatron.core.tensor_parallel.layers' class (:156-167)
            cpl = sys.modules["megatron.core.tensor_parallel.layers"].ColumnParallelLinear
            self.output_layer = cpl(config.hidden_size, vocab_size, config=config, init_method=config.init_method, bias=False,
                                    skip_bias_add=False, gather_output=not parallel_output,
                                    skip_weight_param_allocation=pre_process and share_embeddings_and_output_weights,
                                    embedding_activation_buffer=None, grad_output_buffer=None)
        self.unused = torch.nn.Parameter(0.01 * torch.ones(config.hidden_size))                                         # :180

    def vision_projector_freeze(self):                                                    # :182-191
        for name, param in self.named_parameters():
            if "external_feature_model." in name and ".vit." not in name:
                param.requires_grad = False
        return self

    def vision_model_freeze(self):                                                        # :193-202
        for name, param in self.named_parameters():
            if "external_feature_model.vit." in name:
                param.requires_grad = False
        return self

    def language_model_freeze(self):                                                      # :204-216
        for name, param in self.named_parameters():
            if "external_feature_model." in name or name == "unused":
                continue
            param.requires_grad = False
        return self

    def compute_language_model_loss(self, labels, logits):
        """megatron.core.models.common.language_module.LanguageModule.compute_language_model_loss."""
        labels = labels.transpose(0, 1).contiguous()                                     # [b s] => [s b]
        loss = sys.modules["megatron.core.tensor_parallel"].vocab_parallel_cross_entropy(logits.float(), labels)
        return loss.transpose(0, 1).contiguous()                                         # [s b] => [b s]

    def forward(self, input_ids, position_ids, attention_mask, decoder_input=None, labels=None, inference_params=None,
                packed_seq_params=None, extra_block_kwargs=None, external_inputs={}, tokentype_ids=None, logit_mask=None):
        args = ARGS
        if decoder_input is not None:
            pass
        elif self.pre_process:
            if hasattr(inference_params, "external_inputs") and inference_params.external_inputs is not None \
                    and not inference_params.key_value_memory_dict:                      # :258-262 (first decode step only)
                external_inputs = inference_params.external_inputs
            if external_inputs:                                                           # :264-272
                external_feature = self.external_feature_model(**external_inputs)
                external_feature_dict = {"features": external_feature}
                for k in external_inputs:
                    if "indices" in k or "pre_len" == k:
                        external_feature_dict[k] = external_inputs[k]
                decoder_input = self.embedding(input_ids=input_ids, position_ids=position_ids, external_feature_dict=external_feature_dict)
            else:
                decoder_input = self.embedding(input_ids=input_ids, position_ids=position_ids)
        if hasattr(inference_params, "logit_mask") and inference_params.logit_mask is not None:    # :278-280
            logit_mask = inference_params.logit_mask
        if hasattr(inference_params, "use_kv_cache") and not inference_params.use_kv_cache:        # :282-283
            inference_params = None
        rotary_pos_emb = None
        if self.position_embedding_type == "rope":                                        # :287-295
            rotary_seq_len = self.rotary_pos_emb.get_rotary_seq_len(inference_params, self.decoder, decoder_input, self.config)
            rotary_pos_emb = self.rotary_pos_emb(rotary_seq_len).to(decoder_input.device)
        hidden_states = self.decoder(hidden_states=decoder_input, attention_mask=attention_mask, inference_params=inference_params,
                                     rotary_pos_emb=rotary_pos_emb, packed_seq_params=packed_seq_params, **(extra_block_kwargs or {}))
        hidden_states = hidden_states.clone()                                             # :310-311
        hidden_states += 0.0 * self.unused
        if not self.post_process:
            return hidden_states
        logits, _ = self.output_layer(hidden_states, weight=None, logit_mask=logit_mask)   # :339
        if args.output_multiplier_scale:                                                  # :349-355
            logits = logits * args.output_multiplier_scale
        if args.output_logit_softcapping:
            logits = logits / args.output_logit_softcapping
            logits = torch.tanh(logits)
            logits = logits * args.output_logit_softcapping
        if labels is None:
            return logits.transpose(0, 1).contiguous()                                    # :370
        if logit_mask is not None:                                                        # :372-384
            b = logit_mask.size(0)
            assert b == 1
            with torch.no_grad():
                labels = torch.masked_select(labels, logit_mask).reshape(b, -1)
        if args.is_instruction_dataset:                                                   # :389-391
            labels = labels[:, 1:].contiguous()
            logits = logits[:-1, :, :].contiguous()
        if logits.sum().isnan():
            raise ValueError("found NaN in local forward logits calculation.")
        return self.compute_language_model_loss(labels, logits)                           # :414


def loss_func(loss_mask, output_tensor):
    """M/pretrain_long_vita.py:778-839 without its logging: (sum of the masked losses x CP, number of tokens)."""
    args = ARGS
    losses = output_tensor.float()
    loss_mask = loss_mask[..., 1:].reshape(-1).float() if args.is_instruction_dataset else loss_mask.reshape(-1).float()
    total_tokens = loss_mask.sum()
    loss = torch.cat([torch.sum(losses.view(-1) * loss_mask).view(1), total_tokens.view(1)])
    if args.context_parallel_size > 1:                                                    # :801-803
        from long_vita_amd import parallel_state as own
        # c10d reduces in place and autograd never sees it: the graph of loss[0] stays this rank's own terms
        torch.distributed.all_reduce(loss.detach(), group=own.get_context_parallel_group())
    return loss[0] * args.context_parallel_size, loss[1].clone().detach().to(torch.int)


def forward_step(batch, model, use_logit_mask=True):
    """M/pretrain_long_vita.py:841-869: (tokens, labels, loss_mask, attention_mask, position_ids, external_inputs) -> (output, loss_func)."""
    tokens, labels, loss_mask, attention_mask, position_ids, external_inputs = batch
    logit_mask = loss_mask.bool() if use_logit_mask else None                            # args.logit_mask (:858-860)
    output_tensor = model(tokens, position_ids, attention_mask, labels=labels, external_inputs=external_inputs, logit_mask=logit_mask)
    if use_logit_mask:                                                                    # :866-867
        loss_mask = torch.ones(output_tensor.size(0), output_tensor.size(1) + 1, dtype=output_tensor.dtype, device=output_tensor.device)
    return output_tensor, (lambda out: loss_func(loss_mask, out))


class MegatronVisionModel(torch.nn.Module):
    """The reference's entry-script class (M/pretrain_long_vita.py:310-596), plain torch: what runs when nothing displaces it.
    `args` = the namespace get_args() would return (vision_* flags, image_size, hidden_size); the ViT class, its layer spec and
    MultimodalProjector are taken from the (patched) module names the script imports them from."""

    def __init__(self, args, vit_config, projector_config):
        super().__init__()
        for k in ("vision_seq_length", "image_token_length", "vision_model_type", "vision_context_parallel", "vision_downsample_ratio",
                  "vision_downsample_stride", "add_class_token", "vision_model_freeze", "vision_projector_freeze",
                  "vision_model_recompute", "vision_projector_recompute"):
            setattr(self, k, getattr(args, k))
        vis = "long_vita_megatron.core.models.vision."
        specs = sys.modules[vis + "vit_layer_specs"]
        if self.vision_model_type == "intern_300m":                                       # :346-356 (use_te = False is hard-coded)
            spec, vit_module = specs.get_vit_layer_local_spec_for_intern(), sys.modules[vis + "intern_vit_model"].InternViTModel
        elif self.vision_model_type == "siglip_400m":                                     # :366-372
            spec, vit_module = specs.get_vit_layer_local_spec_for_siglip(), sys.modules[vis + "siglip_vit_model"].SigLIPViTModel
        else:
            raise NotImplementedError(self.vision_model_type)
        self.vit = vit_module(vit_config, spec, add_class_token=args.add_class_token, patch_dim=args.patch_dim, img_h=args.image_size,
                              img_w=args.image_size, vision_context_parallel=args.vision_context_parallel)
        MultimodalProjector = sys.modules[vis + "multimodal_projector"].MultimodalProjector            # imported inside __init__ (:393)
        proj_input_size = vit_config.hidden_size
        if self.vision_downsample_ratio != 1:
            proj_input_size = vit_config.hidden_size * int(1 / self.vision_downsample_ratio) ** 2
        self.vision_projection = MultimodalProjector(projector_config, MLPSubmodules(linear_fc1=None, linear_fc2=None), "mlp",
                                                     proj_input_size)
        self.pre_proj_layernorm = torch.nn.LayerNorm(proj_input_size) if args.vision_projector_pre_norm else torch.nn.Identity()

    def forward_projection(self, vit_output):                                             # :436-450
        return self.vision_projection(self.pre_proj_layernorm(vit_output))

    def forward_downsample(self, vit_output):                                             # :452-470
        if self.add_class_token:
            vit_output = vit_output[:, 1:, :]
        if self.vision_downsample_ratio != 1:
            h = w = int(vit_output.shape[1] ** 0.5)
            vit_output = vit_output.reshape(vit_output.shape[0], h, w, -1)
            vit_output = self.pixel_shuffle(vit_output, scale_factor=self.vision_downsample_ratio)
            vit_output = vit_output.reshape(vit_output.shape[0], -1, vit_output.shape[-1])
        return vit_output

    def forward_once(self, images, attention_mask):                                       # :485-520
        from contextlib import nullcontext
        tp = sys.modules["megatron.core.tensor_parallel"]
        with (torch.no_grad() if self.vision_model_freeze else nullcontext()):
            vit_output = self.vit(images, attention_mask)
            vit_output = tp.checkpoint(self.forward_downsample, False, vit_output) if self.vision_model_recompute else \
                self.forward_downsample(vit_output)
        with (torch.no_grad() if self.vision_projector_freeze else nullcontext()):
            if self.vision_projector_recompute:
                return tp.checkpoint(self.forward_projection, False, vit_output)
            return self.forward_projection(vit_output)

    def forward_chunk(self, images, attention_mask):                                      # :522-533
        return torch.cat([self.forward_once(c, attention_mask) for c in torch.split(images, 256, dim=0)], dim=0)

    def forward(self, **kw_args):                                                         # :535-570 (no vision context parallelism)
        return self.forward_chunk(kw_args["images"], None)

    def pixel_shuffle(self, x, scale_factor=0.5):                                         # :572-582
        n, w, h, c = x.size()
        x = x.view(n, w, int(h * scale_factor), int(c / scale_factor))
        x = x.permute(0, 2, 1, 3).contiguous()
        x = x.view(n, int(h * scale_factor), int(w * scale_factor), int(c / (scale_factor * scale_factor)))
        return x.permute(0, 2, 1, 3).contiguous()


def _vocab_parallel_cross_entropy_placeholder(vocab_parallel_logits, target, label_smoothing=0.0):
    raise AssertionError("Megatron's vocab-parallel cross entropy must not run: the adaptor registers the library's on this name")


_STUBS = {
    "megatron.core.transformer.custom_layers.transformer_engine": dict(TENorm=_TENormPlaceholder),
    "megatron.core.transformer.transformer_block": dict(TransformerBlock=TransformerBlock, TENorm=_TENormPlaceholder),
    "megatron.core.tensor_parallel": dict(checkpoint=checkpoint, vocab_parallel_cross_entropy=_vocab_parallel_cross_entropy_placeholder),
    "megatron.core.tensor_parallel.random": dict(checkpoint=checkpoint),      # defined there, re-exported by the package (as in Megatron)
    "megatron.core.tensor_parallel.cross_entropy": dict(vocab_parallel_cross_entropy=_vocab_parallel_cross_entropy_placeholder),
    "long_vita_megatron.core.models.multimodal.gpt_vl_model": dict(GPTVLModel=GPTVLModel),
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
                                "long_vita_megatron.core.models.vision.vit_layer_specs", "long_vita_megatron.core.models.vision.intern_vit_model",
                                "long_vita_megatron.core.models.vision.siglip_vit_model",
                                "long_vita_megatron.core.models.vision.multimodal_projector"]:
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
