"""Tensor-parallel linears and norms of the decoder layer as Megatron-constructible, HIP-backed `torch.nn.Module`s.

Mirrors (constructor signatures, parameter names, `(output, bias)` returns, error behaviour):
  ColumnParallelLinear            M/core/tensor_parallel/layers.py:650-904 (forward with `logit_mask` :825-904)
  RowParallelLinear               M/core/tensor_parallel/layers.py:921-1115
  LayerNormColumnParallelLinear   TELayerNormColumnParallelLinear as the TE layer spec uses it
                                  (M/core/models/gpt/gpt_layer_specs.py:39,93-104; parameters `layer_norm_weight`, `weight`, `bias`
                                  — the names M/ckpt_convert_modellink_to_megatron_with_te.py:37-41 converts to)
  Norm                            PTNorm / TENorm (M/core/transformer/custom_layers/transformer_engine.py:13-51): RMSNorm | LayerNorm
so `ModuleSpec` / `build_module` construct them exactly as they construct Megatron's own, `state_dict()` carries
Megatron's keys, and every forward / backward is a libvita_hip.so kernel through autograd_fns.  `--sequence-parallel`
(layers.py:392-399,483-494,1095): the column-parallel linears all-gather their input along the sequence over the
tensor-parallel group (RCCL), the row-parallel linears reduce-scatter their output.
"""
from __future__ import annotations

import os
from typing import Callable, Optional

import torch
from torch.nn import Parameter

from . import autograd_fns as F_, ops, parallel_state as mpu


# One autograd node per fused module (NormLinearFn, GatedMLPFn) instead of one per kernel: the same kernels in the same order, but the
# normed rows and the gated activation are re-derived in the backward instead of kept (0.8 GB less per 16K layer).  VITA_MODULE_UNFUSED=1
# restores the node-per-kernel graph (A / B and bit-identity tests).
FUSE_AUTOGRAD_NODES = os.environ.get("VITA_MODULE_UNFUSED", "0") in ("", "0")


def _divide(n: int, d: int) -> int:
    if n % d:
        raise ValueError(f"{n} is not divisible by {d}")
    return n // d


def _alloc(config, *shape) -> torch.Tensor:
    """Parameters live on the current HIP device unless `config.use_cpu_initialization` (layers.py:744-775)."""
    dtype = getattr(config, "params_dtype", torch.bfloat16)
    if getattr(config, "use_cpu_initialization", False):
        return torch.empty(*shape, dtype=dtype)
    return torch.empty(*shape, dtype=dtype, device=torch.cuda.current_device())


def _add_extra_state_hook(module: torch.nn.Module) -> None:
    """layers.py:819-823: checkpoints written without TE have no `_extra_state` entry."""
    module._register_load_state_dict_pre_hook(
        lambda state_dict, prefix, *args, **kwargs: state_dict.setdefault(f"{prefix}_extra_state"))
    # a checkpoint load rewrites frozen weights too: drop the zero-padded copies ops.gemm keeps of them (`copy_` bumps torch's
    # version counter, loaders that assign through `.data` do not)
    module.register_load_state_dict_post_hook(lambda m, incompatible_keys: ops.invalidate_padded_weights())


class _TEStateMixin:
    def set_extra_state(self, state):
        """Extra state is ignored (layers.py:911-912)."""

    def get_extra_state(self):
        """Keep compatibility with TE state dicts (layers.py:914-916)."""
        return None


class ColumnParallelLinear(_TEStateMixin, torch.nn.Module):
    """Y = X A^T + b with A split along its rows (output features) over the tensor-parallel group."""

    def __init__(self, input_size, output_size, *, config, init_method: Optional[Callable], bias=True, gather_output=False,
                 stride=1, keep_master_weight_for_test=False, skip_bias_add=False, skip_weight_param_allocation: bool = False,
                 embedding_activation_buffer=None, grad_output_buffer=None, is_expert: bool = False,
                 tp_comm_buffer_name: str = None, disable_grad_reduce: bool = False):
        super().__init__()
        if is_expert or embedding_activation_buffer is not None or grad_output_buffer is not None:
            raise NotImplementedError("MoE experts / deferred embedding wgrad are not on the Long-VITA path")
        self.input_size, self.output_size = input_size, output_size
        self.gather_output, self.skip_bias_add, self.config = gather_output, skip_bias_add, config
        self.disable_grad_reduce = disable_grad_reduce
        world = mpu.get_tensor_model_parallel_world_size()
        self.output_size_per_partition = _divide(output_size, world)
        if not skip_weight_param_allocation:
            self.weight = Parameter(_alloc(config, self.output_size_per_partition, input_size))
            if getattr(config, "perform_initialization", True) and init_method is not None:
                with torch.no_grad():
                    init_method(self.weight)
            setattr(self.weight, "allreduce", True)
            setattr(self.weight, "tensor_model_parallel", True)
            setattr(self.weight, "partition_dim", 0)
            setattr(self.weight, "partition_stride", stride)
        else:
            self.weight = None
        if bias:
            self.bias = Parameter(_alloc(config, self.output_size_per_partition))
            with torch.no_grad():
                self.bias.zero_()                                                    # "Always initialize bias to zero."
            setattr(self.bias, "allreduce", True)
            setattr(self.bias, "tensor_model_parallel", True)
            setattr(self.bias, "partition_dim", 0)
            setattr(self.bias, "partition_stride", stride)
        else:
            self.register_parameter("bias", None)
        self.sequence_parallel = bool(getattr(config, "sequence_parallel", False)) and world > 1   # :790-796
        self.allreduce_dgrad = world > 1 and not self.sequence_parallel
        if getattr(config, "gradient_accumulation_fusion", False):
            raise RuntimeError("gradient_accumulation_fusion needs APEX's fused_weight_gradient_mlp_cuda, which does not exist "
                               "on this platform; run without it")                   # :800-811
        _add_extra_state_hook(self)

    @classmethod
    def from_weight(cls, weight: Optional[torch.Tensor], bias: Optional[torch.Tensor] = None, skip_bias_add: bool = False):
        """Stand-alone driver (gpt_vl_model.GPTVLModel): wrap existing (frozen, inference) tensors."""
        self = cls.__new__(cls)
        torch.nn.Module.__init__(self)
        self.input_size = None if weight is None else weight.shape[1]
        self.output_size_per_partition = None if weight is None else weight.shape[0]
        self.output_size = None
        self.gather_output, self.skip_bias_add, self.config = False, skip_bias_add, None
        self.disable_grad_reduce, self.sequence_parallel, self.allreduce_dgrad = False, False, False
        self.weight = None if weight is None else Parameter(weight, requires_grad=False)
        if bias is None:
            self.register_parameter("bias", None)
        else:
            self.bias = Parameter(bias, requires_grad=False)
        return self

    def forward(self, input_: torch.Tensor, weight: Optional[torch.Tensor] = None, logit_mask=None):
        """input_ [s, b, hidden] -> (output [n_sel or s, b, out / TP], bias or None)  (layers.py:825-904)."""
        if weight is None:
            if self.weight is None:
                raise RuntimeError("weight was not supplied to ColumnParallelLinear forward pass "
                                   "and skip_weight_param_allocation is True.")
            weight = self.weight
        elif self.output_size_per_partition is not None:
            expected_shape = (self.output_size_per_partition, self.input_size)
            if tuple(weight.shape) != expected_shape:
                raise RuntimeError(f"supplied weight's shape is {tuple(weight.shape)}, not {expected_shape} as expected")
        return self._linear(input_, weight, logit_mask, fuse_bias=not self.skip_bias_add)

    def _linear(self, input_, weight, logit_mask, fuse_bias: bool):
        """fuse_bias: the bias is added inside the GEMM epilogue (one rounding); else it is handed back beside the output."""
        bias = self.bias if fuse_bias else None
        if self.allreduce_dgrad or self.sequence_parallel or self.disable_grad_reduce:
            input_parallel = input_
        else:
            input_parallel = F_.CopyToTP.apply(input_)                               # :872
        output_parallel = F_.LinearFn.apply(input_parallel, weight, bias, self.allreduce_dgrad, self.sequence_parallel,
                                            logit_mask)
        if self.gather_output:
            assert not self.sequence_parallel                                         # :897
            output = F_.GatherFromTP.apply(output_parallel)
        else:
            output = output_parallel
        return output, (None if fuse_bias else self.bias)

    def forward_fused_bias(self, input_: torch.Tensor):
        """The same linear with its bias folded into the GEMM epilogue whatever `skip_bias_add` says (ViTMLP: Megatron's MLP adds the
        fc1 bias itself as a torch op; here it stays in the kernel)."""
        return self._linear(input_, self.weight, None, fuse_bias=True)


class RowParallelLinear(_TEStateMixin, torch.nn.Module):
    """Y = X A^T + b with A split along its columns (input features); partial sums are reduced over the TP group."""

    def __init__(self, input_size: int, output_size: int, *, config, init_method: Optional[Callable], bias: bool,
                 input_is_parallel: bool, skip_bias_add: bool, stride: int = 1, keep_master_weight_for_test: bool = False,
                 is_expert: bool = False, tp_comm_buffer_name: str = None):
        super().__init__()
        if is_expert:
            raise NotImplementedError("MoE experts are not on the Long-VITA path")
        self.input_size, self.output_size = input_size, output_size
        self.input_is_parallel, self.skip_bias_add, self.config = input_is_parallel, skip_bias_add, config
        world = mpu.get_tensor_model_parallel_world_size()
        self.sequence_parallel = bool(getattr(config, "sequence_parallel", False)) and world > 1
        if getattr(config, "sequence_parallel", False) and not input_is_parallel:
            raise RuntimeError("To enable `sequence_parallel`, `input_is_parallel` must be `True`")   # :968-969
        if not input_is_parallel and world > 1:
            raise NotImplementedError("scatter_to_tensor_model_parallel_region (input_is_parallel=False) is not on this path")
        self.input_size_per_partition = _divide(input_size, world)
        self.weight = Parameter(_alloc(config, output_size, self.input_size_per_partition))
        if getattr(config, "perform_initialization", True) and init_method is not None:
            with torch.no_grad():
                init_method(self.weight)
        setattr(self.weight, "allreduce", True)
        setattr(self.weight, "tensor_model_parallel", True)
        setattr(self.weight, "partition_dim", 1)
        setattr(self.weight, "partition_stride", stride)
        if bias:
            self.bias = Parameter(_alloc(config, output_size))
            with torch.no_grad():
                self.bias.zero_()
            setattr(self.bias, "allreduce", True)
            setattr(self.bias, "sequence_parallel", self.sequence_parallel)
        else:
            self.register_parameter("bias", None)
        if getattr(config, "gradient_accumulation_fusion", False):
            raise RuntimeError("gradient_accumulation_fusion needs APEX's fused_weight_gradient_mlp_cuda, which does not exist "
                               "on this platform; run without it")
        _add_extra_state_hook(self)

    def forward(self, input_: torch.Tensor):
        """input_ [s, b, in / TP] -> (output [s (/ TP with sequence parallelism), b, out], bias or None)  (:1059-1115)."""
        output_parallel = F_.LinearFn.apply(input_, self.weight, None, False, False, None)
        if self.sequence_parallel:
            output_ = F_.ReduceScatterToSP.apply(output_parallel)                    # :1095
        else:
            output_ = F_.ReduceFromTP.apply(output_parallel)                         # :1097
        if not self.skip_bias_add:
            if self.bias is not None:                                                 # not on the Long-VITA path (no linear bias)
                s, b, n = output_.shape
                output_ = BiasAddFn.apply(output_, self.bias)
            return output_, None
        return output_, self.bias


class BiasAddFn(torch.autograd.Function):
    """output + bias behind a row-parallel reduction (layers.py:1099): vita_gemm is not involved, so the add is the
    library's bias epilogue applied through an identity-free path: y = x + b rounded once to bf16."""

    @staticmethod
    def forward(ctx, x, bias):
        rows = x.numel() // x.shape[-1]
        return ops.add(x.contiguous().view(rows, -1), bias.expand(rows, -1).contiguous()).view_as(x)      # one pass, x is not copied

    @staticmethod
    def backward(ctx, g):
        g2 = g.reshape(-1, g.shape[-1]).contiguous()
        return g, F_.bias_grad(g2)


class RMSNorm(torch.nn.Module):
    """M/core/transformer/custom_layers/transformer_engine.py:54-79 (same constructor and `weight` parameter)."""

    def __init__(self, dim: int, eps: float = 1e-6, sequence_parallel: bool = False, config=None):
        super().__init__()
        self.eps = eps
        self.weight = Parameter(torch.ones(dim) if config is None else torch.ones_like(_alloc(config, dim)))
        setattr(self.weight, "sequence_parallel", sequence_parallel)

    def forward(self, x):
        return F_.RMSNormFn.apply(x, self.weight, self.eps)


class LayerNorm(torch.nn.Module):
    """torch.nn.LayerNorm's parameters (`weight`, `bias`); forward and backward are library kernels (the ViT block norms:
    TENorm with normalization = "LayerNorm", M/pretrain_long_vita.py:216)."""

    def __init__(self, normalized_shape: int, eps: float = 1e-5, config=None):
        super().__init__()
        self.eps = eps
        self.weight = Parameter(torch.ones(normalized_shape) if config is None else torch.ones_like(_alloc(config, normalized_shape)))
        self.bias = Parameter(torch.zeros_like(self.weight))

    def forward(self, x):
        if torch.is_grad_enabled() and (x.requires_grad or self.weight.requires_grad):
            return F_.LayerNormFn.apply(x, self.weight, self.bias, self.eps)       # vita_layernorm_fwd / vita_layernorm_bwd
        return ops.layernorm(x, self.weight, self.bias, self.eps)


class Norm:
    """PTNorm / TENorm: `Norm(config=..., hidden_size=..., eps=...)` returns the instance the config asks for (:13-51)."""

    def __new__(cls, config, hidden_size: int, eps: float = 1e-5):
        if config.normalization == "LayerNorm":
            return LayerNorm(hidden_size, eps=eps, config=config)
        if config.normalization == "RMSNorm":
            return RMSNorm(dim=hidden_size, eps=eps, sequence_parallel=getattr(config, "sequence_parallel", False), config=config)
        raise Exception("Only LayerNorm and RMSNorm are curently supported")


class LayerNormColumnParallelLinear(ColumnParallelLinear):
    """norm(x) -> column-parallel linear in one module (TELayerNormColumnParallelLinear): vita_rmsnorm_fwd -> vita_gemm_bf16.
    TE's flat parameter names: `layer_norm_weight` (`layer_norm_bias`) beside `weight`, `bias`.  Under sequence parallelism
    the norm runs on the rank's sequence shard and the normed rows are all-gathered inside the linear (TE's order)."""

    def __init__(self, input_size: int, output_size: int, *, config, init_method: Optional[Callable], gather_output: bool,
                 bias: bool, skip_bias_add: bool, is_expert: bool, skip_weight_param_allocation: bool = False,
                 tp_comm_buffer_name: str = None):
        if gather_output:
            raise ValueError("Transformer Engine linear layers do not support gather_output = True")
        if is_expert:
            raise ValueError("Transformer Engine linear layers do not yet support MoE")
        super().__init__(input_size, output_size, config=config, init_method=init_method, bias=bias, gather_output=False,
                         skip_bias_add=skip_bias_add, skip_weight_param_allocation=skip_weight_param_allocation)
        self.eps = config.layernorm_epsilon
        self.normalization = getattr(config, "normalization", "RMSNorm")
        self.layer_norm_weight = Parameter(torch.ones_like(_alloc(config, input_size)))
        setattr(self.layer_norm_weight, "sequence_parallel", getattr(config, "sequence_parallel", False))
        if self.normalization == "LayerNorm":
            self.layer_norm_bias = Parameter(torch.zeros_like(self.layer_norm_weight))
            setattr(self.layer_norm_bias, "sequence_parallel", getattr(config, "sequence_parallel", False))
        else:
            self.register_parameter("layer_norm_bias", None)

    def _norm(self, x: torch.Tensor) -> torch.Tensor:
        if self.normalization == "RMSNorm":
            return F_.RMSNormFn.apply(x, self.layer_norm_weight, self.eps)
        if torch.is_grad_enabled() and (x.requires_grad or self.layer_norm_weight.requires_grad):
            return F_.LayerNormFn.apply(x, self.layer_norm_weight, self.layer_norm_bias, self.eps)
        return ops.layernorm(x, self.layer_norm_weight, self.layer_norm_bias, self.eps)

    def _one_node(self, x: torch.Tensor) -> bool:
        """RMSNorm + linear as one autograd node (NormLinearFn keeps x, not the normed copy): the decoder's qkv / fc1 under autograd."""
        return (FUSE_AUTOGRAD_NODES and self.normalization == "RMSNorm" and self.weight is not None and not self.disable_grad_reduce
                and torch.is_grad_enabled()
                and (x.requires_grad or self.weight.requires_grad or self.layer_norm_weight.requires_grad))

    def forward(self, x: torch.Tensor):
        if self._one_node(x):
            fuse_bias = not self.skip_bias_add
            out = F_.NormLinearFn.apply(x, self.layer_norm_weight, self.weight, self.bias if fuse_bias else None, self.eps,
                                        self.allreduce_dgrad, self.sequence_parallel)
            return out, (None if fuse_bias else self.bias)
        return super().forward(self._norm(x))

    def forward_fused_bias(self, x: torch.Tensor):
        return super().forward_fused_bias(self._norm(x))


class ViTMLP(torch.nn.Module):
    """The ViT's dense MLP with Megatron MLP's constructor (`MLP(config, submodules, is_expert=False, input_size=None)`,
    megatron/core/transformer/mlp.py) and return value `(output, output_bias)`: linear_fc1 (+ bias) -> GELU -> linear_fc2.
    Megatron's MLP.forward adds the fc1 bias and applies `config.activation_func` (torch.nn.functional.gelu,
    M/pretrain_long_vita.py:207) as two torch ops; here, without autograd, bias + GELU are the epilogue of the fc1 GEMM
    (VITA_EPI_BIAS_GELU: one kernel), and with autograd the GEMM (+ bias) is followed by vita_gelu_fwd, whose input the backward keeps."""

    def __init__(self, config, submodules, is_expert: bool = False, input_size: int = None, unfused_bias: bool = False):
        super().__init__()
        import functools
        from megatron.core.transformer.spec_utils import build_module
        if is_expert or getattr(config, "gated_linear_unit", False):
            raise NotImplementedError("ViTMLP is the dense, non-gated MLP of the vision encoders")
        act = getattr(config, "activation_func", torch.nn.functional.gelu)
        # InternViT: torch.nn.functional.gelu (M/pretrain_long_vita.py:207); SigLIP: partial(F.gelu, approximate="tanh") (:290)
        self.tanh = (isinstance(act, functools.partial) and act.func is torch.nn.functional.gelu and not act.args
                     and act.keywords == {"approximate": "tanh"})
        if act is not torch.nn.functional.gelu and not self.tanh:
            raise NotImplementedError("ViTMLP: activation_func must be F.gelu or partial(F.gelu, approximate='tanh')")
        # unfused_bias (every local ViT spec of the reference: `bias_activation_fusion = False`, M/pretrain_long_vita.py:213,296): Megatron's
        # MLP adds the fc1 bias to the bf16-ROUNDED product as an op of its own before the activation (skip_bias_add linears); False = the
        # TransformerEngine order (bias inside the GEMM, one rounding) of the TE spec
        self.unfused_bias = bool(unfused_bias)
        self.config = config
        self.input_size = input_size if input_size is not None else config.hidden_size
        self.linear_fc1 = build_module(submodules.linear_fc1, self.input_size, config.ffn_hidden_size, config=config,
                                       init_method=config.init_method, gather_output=False, bias=config.add_bias_linear,
                                       skip_bias_add=True, is_expert=False, tp_comm_buffer_name="fc1")
        self.activation_func = act
        self.linear_fc2 = build_module(submodules.linear_fc2, config.ffn_hidden_size, config.hidden_size, config=config,
                                       init_method=config.output_layer_init_method, bias=config.add_bias_linear, input_is_parallel=True,
                                       skip_bias_add=True, is_expert=False, tp_comm_buffer_name="fc2")

    def forward(self, hidden_states):
        fc1 = self.linear_fc1
        grad = torch.is_grad_enabled() and (hidden_states.requires_grad or fc1.weight.requires_grad)
        plain = type(fc1) is ColumnParallelLinear and not fc1.sequence_parallel and mpu.get_tensor_model_parallel_world_size() == 1
        fused_epi = {(False, False): ops.EPI_BIAS_GELU, (True, True): ops.EPI_BIAS2_GELU_TANH,
                     (True, False): ops.EPI_BIAS2_GELU}.get((self.unfused_bias, self.tanh))
        if fc1.bias is None:                                                  # the projector's MLP (add_bias_linear = False): gelu(bf16(x W^T))
            fused_epi = None if self.tanh else ops.EPI_BIAS_GELU
        if not grad and plain and fused_epi is not None:
            s, b, h = hidden_states.shape                                     # inference: GEMM + bias + GELU in one kernel
            a = ops.gemm(hidden_states.reshape(s * b, h), fc1.weight, fused_epi, fc1.bias).view(s, b, -1)
        elif self.unfused_bias:
            y, bias = fc1(hidden_states)                                      # skip_bias_add: the bf16 product and the bias, added as an op of its own
            a = F_.GeluFn.apply(y if bias is None else BiasAddFn.apply(y, bias), self.tanh)
        else:
            y, _ = fc1.forward_fused_bias(hidden_states)                      # GEMM + bias (one rounding), pre-activation kept for the backward
            a = F_.GeluFn.apply(y, self.tanh)
        return self.linear_fc2(a)                                              # (output, bias): the bias goes into the residual kernel


class GatedMLP(torch.nn.Module):
    """The decoder's SwiGLU MLP with Megatron MLP's constructor and `(output, output_bias)` return (megatron/core/transformer/mlp.py;
    stage-3 flags `--swiglu --disable-bias-linear`).  Megatron's MLP.forward runs linear_fc1, then `silu(gate) * up` as torch ops on
    the [s, b, 2 ffn] product; here, without autograd, the gated activation is the EPILOGUE of the fc1 GEMM (VITA_EPI_SWIGLU: the
    2 ffn-wide product never reaches HBM — what GPTVLModel.decoder_layer does), and with autograd fc1 is followed by vita_swiglu_fwd
    (SwiGLUFn keeps the product for the backward).  fc1's weight rows are cat[gate, up] (R/tools/hf2mcore_long_vita.py:612)."""

    def __init__(self, config, submodules, is_expert: bool = False, input_size: int = None):
        super().__init__()
        from megatron.core.transformer.spec_utils import build_module
        if is_expert or not getattr(config, "gated_linear_unit", False) or getattr(config, "add_bias_linear", False):
            raise NotImplementedError("GatedMLP is the dense bias-free SwiGLU MLP of the Long-VITA decoder")
        if getattr(config, "activation_func", torch.nn.functional.silu) is not torch.nn.functional.silu:
            raise NotImplementedError("GatedMLP: activation_func must be silu (--swiglu)")
        self.config = config
        self.input_size = input_size if input_size is not None else config.hidden_size
        self.linear_fc1 = build_module(submodules.linear_fc1, self.input_size, 2 * config.ffn_hidden_size, config=config,
                                       init_method=config.init_method, gather_output=False, bias=False, skip_bias_add=True,
                                       is_expert=False, tp_comm_buffer_name="fc1")
        self.activation_func = config.activation_func
        self.linear_fc2 = build_module(submodules.linear_fc2, config.ffn_hidden_size, config.hidden_size, config=config,
                                       init_method=config.output_layer_init_method, bias=False, input_is_parallel=True,
                                       skip_bias_add=True, is_expert=False, tp_comm_buffer_name="fc2")

    def forward(self, hidden_states):
        fc1 = self.linear_fc1
        grad = torch.is_grad_enabled() and (hidden_states.requires_grad or fc1.weight.requires_grad)
        fusable = not grad and not fc1.sequence_parallel and mpu.get_tensor_model_parallel_world_size() == 1
        if fusable:
            x = fc1._norm(hidden_states) if isinstance(fc1, LayerNormColumnParallelLinear) else hidden_states
            s, b, h = x.shape
            a = ops.gemm(x.reshape(s * b, h), fc1.weight, ops.EPI_SWIGLU).view(s, b, -1)
        elif FUSE_AUTOGRAD_NODES and not (isinstance(fc1, LayerNormColumnParallelLinear) and fc1.normalization != "RMSNorm"):
            # one autograd node for [norm ->] fc1 -> SwiGLU -> fc2 [-> TP reduction]: keeps x and the fc1 product only (GatedMLPFn)
            ln_w = fc1.layer_norm_weight if isinstance(fc1, LayerNormColumnParallelLinear) else None
            out = F_.GatedMLPFn.apply(hidden_states, ln_w, fc1.weight, self.linear_fc2.weight,
                                      getattr(fc1, "eps", 0.0), fc1.sequence_parallel)
            return out, None
        else:
            y, _ = fc1(hidden_states)
            a = F_.SwiGLUFn.apply(y)
        return self.linear_fc2(a)


class ResidualAddFn(torch.autograd.Function):
    """residual + x as one library kernel (vita_add_bf16); the gradient passes to both unchanged."""

    @staticmethod
    def forward(ctx, x, residual):
        return ops.add(x.contiguous(), residual.contiguous())            # out = bf16(x + residual): one pass, nothing is cloned

    @staticmethod
    def backward(ctx, g):
        return g, g


def get_bias_dropout_add(training: bool, fused: bool):
    """megatron.core.fusions.fused_bias_dropout.get_bias_dropout_add for this path: no linear bias, dropout 0 (stage-3
    `--attention-dropout 0.0 --hidden-dropout 0.0`): residual + x through the library (ResidualAddFn); a bias or dropout > 0
    in training mode takes Megatron's unfused torch expression (not on the Long-VITA path)."""
    def _bda(x_with_bias, residual, prob):
        x, bias = x_with_bias
        if bias is None and (prob == 0.0 or not training):
            if not (torch.is_grad_enabled() and (x.requires_grad or residual.requires_grad)) and x.is_contiguous():
                return ops.add(x, residual.contiguous(), out=x)          # inference: into the linear's own (temporary) output
            return ResidualAddFn.apply(x, residual)
        if bias is not None:
            x = x + bias
        out = torch.nn.functional.dropout(x, p=prob, training=training)
        return residual + out

    return _bda
