"""Plain-torch LEAF modules for running the reference's own composite classes on CPU.  TEST INFRASTRUCTURE ONLY.

`oracle/make_golden.py` imports the reference's real `GPTVLModel` (M/core/models/multimodal/gpt_vl_model.py:72-416), `TransformerBlock`
(M/core/transformer/transformer_block.py:119-420), `InternViTModel` (M/core/models/vision/intern_vit_model.py:92-261) and executes the
source of `MegatronVisionModel` / `forward_step` (M/pretrain_long_vita.py:310-596, 841-869) with the classes below bound to the names
those files import from Megatron-LM (absent here: a git submodule the reference does not vendor, R/.gitmodules:4-6).  The leaves carry
no reference logic of their own — each is the smallest deterministic fp32 module with the constructor / call signature the composite
uses — so what the resulting fixtures (`tests/golden/gptvl_forward.pt`, `vision_model.pt`, `transformer_block.pt`, `intern_vit_forward.pt`)
pin is the COMPOSITION the reference wrote: which leaf is called with what, the logit-mask / instruction-shift / soft-cap tail, the
freeze and recompute contexts, the chunking.  `tests/test_oracle_golden.py` builds `tests/dummy_megatron.py`'s restatements over the
same leaves and requires them to reproduce the fixtures bit for bit.

Two leaves restate published Megatron-LM behaviour (core_r0.7.0) because the composites inherit it: `LanguageModule`
(compute_language_model_loss: [b s] -> [s b], vocab_parallel_cross_entropy(logits.float(), labels), back to [b s]) and
`checkpoint` (tensor_parallel.checkpoint = re-entrant activation checkpointing)."""
from __future__ import annotations

import dataclasses
import types
import zlib

import torch


def init_by_name(module: torch.nn.Module, seed: int = 0, scale: float = 0.3) -> None:
    """Every parameter <- N(0, scale) from a generator seeded by (seed, crc32(parameter name)): two classes with the same parameter NAMES
    get the same values whatever their construction order."""
    with torch.no_grad():
        for name, p in module.named_parameters():
            g = torch.Generator().manual_seed((seed * 1000003 + zlib.crc32(name.encode())) % (2 ** 31))
            p.copy_(torch.randn(p.shape, generator=g) * scale)


def config(**kw):
    base = dict(hidden_size=16, kv_channels=8, rotary_interleaved=False, defer_embedding_wgrad_compute=False, context_parallel_size=1,
                init_method=None, num_layers=3, layernorm_epsilon=1e-5, recompute_granularity=None, recompute_method=None,
                recompute_num_layers=None, distribute_saved_activations=False, fp8=None, cpu_offloading=False, sequence_parallel=False,
                enable_cuda_graph=False, independent_parallel=True, pipeline_model_parallel_size=1, normalization="LayerNorm")
    base.update(kw)
    return types.SimpleNamespace(**base)


class MegatronModule(torch.nn.Module):
    def __init__(self, config=None):
        super().__init__()
        self.config = config


def vocab_parallel_cross_entropy(vocab_parallel_logits, target, label_smoothing=0.0):
    """TP = 1, in-range targets: the per-token cross entropy [s, b] fp32."""
    s, b, v = vocab_parallel_logits.shape
    return torch.nn.functional.cross_entropy(vocab_parallel_logits.reshape(s * b, v), target.reshape(s * b), reduction="none").view(s, b)


class LanguageModule(MegatronModule):
    """megatron.core.models.common.language_module.language_module.LanguageModule (published): the three members GPTVLModel uses."""

    def compute_language_model_loss(self, labels, logits):
        labels = labels.transpose(0, 1).contiguous()                       # [b s] => [s b]
        loss = vocab_parallel_cross_entropy(logits.float(), labels)
        return loss.transpose(0, 1).contiguous()                           # [s b] => [b s]

    def setup_embeddings_and_output_layer(self):                           # PP = 1, untied weights: nothing to link
        pass

    def shared_embedding_or_output_weight(self):
        return self.embedding.word_embeddings.weight


class Embedding(torch.nn.Module):
    """LanguageModelEmbedding's signature: (input_ids [b, s], position_ids, external_feature_dict) -> [s, b, h]."""

    def __init__(self, config, vocab_size, max_sequence_length, position_embedding_type="learned_absolute"):
        super().__init__()
        self.word_embeddings = torch.nn.Embedding(vocab_size, config.hidden_size)

    def forward(self, input_ids, position_ids, external_feature_dict=None):
        emb = self.word_embeddings(input_ids)                              # [b, s, h]
        if external_feature_dict:
            feats = external_feature_dict["features"].reshape(-1, emb.shape[-1])
            idx = external_feature_dict["indices"]
            emb = emb.clone()
            emb[0, idx] = feats[:idx.numel()]
            if "pre_len" in external_feature_dict:                         # proves the key travelled
                emb = emb + 1e-3 * float(external_feature_dict["pre_len"])
        return emb.transpose(0, 1).contiguous()


class Rotary(torch.nn.Module):
    def __init__(self, kv_channels, rotary_percent=1.0, rotary_interleaved=False, seq_len_interpolation_factor=None, rotary_base=10000):
        super().__init__()
        self.kv_channels, self.rotary_base = kv_channels, rotary_base

    def get_rotary_seq_len(self, inference_params, transformer, transformer_input, transformer_config):
        return transformer_input.size(0) * transformer_config.context_parallel_size

    def forward(self, max_seq_len, offset=0):
        return (torch.arange(max_seq_len, dtype=torch.float32) / self.rotary_base)[:, None, None, None].expand(-1, 1, 1, self.kv_channels)


class Block(torch.nn.Module):
    """TransformerBlock's signature as GPTVLModel / InternViTModel use it."""

    def __init__(self, config, spec=None, post_layer_norm=True, pre_process=True, post_process=True):
        super().__init__()
        self.lin = torch.nn.Linear(config.hidden_size, config.hidden_size)
        self.input_tensor = None
        self.seen = {}

    def set_input_tensor(self, t):
        self.input_tensor = t

    def forward(self, hidden_states, attention_mask=None, context=None, context_mask=None, rotary_pos_emb=None, inference_params=None,
                packed_seq_params=None):
        self.seen = dict(inference_params=inference_params, attention_mask=attention_mask, packed_seq_params=packed_seq_params)
        y = torch.tanh(self.lin(hidden_states)) + hidden_states
        if rotary_pos_emb is not None:
            y = y + rotary_pos_emb[:y.size(0), :, 0, :1]
        return y


class ColumnParallelLinear(torch.nn.Module):
    """The output layer as GPTVLModel builds and calls it: (hidden [s, b, c], weight=None, logit_mask=[b, s] bool) -> (logits, None); with
    a mask only the selected rows reach the GEMM (M/core/tensor_parallel/layers.py, the reference's masked forward)."""

    def __init__(self, input_size, output_size, *, config, init_method, bias=True, gather_output=False, skip_bias_add=False,
                 skip_weight_param_allocation=False, embedding_activation_buffer=None, grad_output_buffer=None, **kw):
        super().__init__()
        self.built_with = dict(bias=bias, gather_output=gather_output, skip_bias_add=skip_bias_add,
                               skip_weight_param_allocation=skip_weight_param_allocation)
        self.weight = torch.nn.Parameter(torch.zeros(output_size, input_size))

    def forward(self, input_, weight=None, logit_mask=None):
        w = self.weight if weight is None else weight
        if logit_mask is not None:
            s, b, c = input_.shape
            input_ = torch.masked_select(input_, logit_mask.transpose(0, 1).unsqueeze(2)).reshape(-1, b, c)
        return torch.matmul(input_, w.t()), None


class FeatureModel(torch.nn.Module):
    """An external feature model with the parameter-name structure the freeze methods look at: `vit.*` and everything else."""

    def __init__(self, config, *external_args):
        super().__init__()
        self.external_args = external_args
        self.vit = torch.nn.Linear(12, config.hidden_size)
        self.vision_projection = torch.nn.Linear(config.hidden_size, config.hidden_size)

    def forward(self, **kw):
        return self.vision_projection(torch.tanh(self.vit(kw["images"].reshape(kw["images"].shape[0], -1, 12))))


def checkpoint(function, distribute_saved_activations, *args):
    """megatron.core.tensor_parallel.checkpoint: re-entrant activation checkpointing (forward under no_grad, re-run in backward)."""
    from torch.utils.checkpoint import checkpoint as ckpt
    return ckpt(function, *args, use_reentrant=True)


class ViT(torch.nn.Module):
    """A vision tower with InternViTModel's constructor keywords: images [n, 3, H, W] -> [n, grid^2 (+ 1), h]."""

    def __init__(self, config, spec, add_class_token=True, class_token_len=1, patch_dim=14, img_h=336, img_w=336, vision_context_parallel=False):
        super().__init__()
        self.built_with = dict(spec=spec, add_class_token=add_class_token, patch_dim=patch_dim, img_h=img_h, img_w=img_w,
                               vision_context_parallel=vision_context_parallel, hidden_size=config.hidden_size)
        self.grid, self.patch, self.cls = img_h // patch_dim, patch_dim, add_class_token
        self.proj = torch.nn.Linear(3 * patch_dim * patch_dim, config.hidden_size)
        self.class_token = torch.nn.Parameter(torch.zeros(1, 1, config.hidden_size))

    def forward(self, images, attention_mask=None):
        n, p, g = images.shape[0], self.patch, self.grid
        x = images.reshape(n, 3, g, p, g, p).permute(0, 2, 4, 1, 3, 5).reshape(n, g * g, 3 * p * p)
        x = torch.tanh(self.proj(x))
        if self.cls:
            x = torch.cat([self.class_token.expand(n, -1, -1), x], dim=1)
        return x


class Projector(torch.nn.Module):
    """MultimodalProjector's constructor: (config, submodules, projector_type, input_size)."""

    def __init__(self, config, submodules, projector_type, input_size):
        super().__init__()
        self.built_with = dict(projector_type=projector_type, input_size=input_size, hidden_size=config.hidden_size,
                               ffn_hidden_size=config.ffn_hidden_size, gated_linear_unit=config.gated_linear_unit,
                               add_bias_linear=config.add_bias_linear, bias_activation_fusion=config.bias_activation_fusion,
                               activation_func=getattr(config.activation_func, "__name__", str(config.activation_func)))
        self.lin = torch.nn.Linear(input_size, config.hidden_size)

    def forward(self, x):
        return self.lin(x)


class BaseTransformerLayer:
    """megatron.core.transformer.transformer_layer.BaseTransformerLayer: the marker class TransformerBlock's spec dispatch tests."""


@dataclasses.dataclass
class ModuleSpec:
    """megatron.core.transformer.spec_utils.ModuleSpec (published)."""
    module: object
    params: dict = dataclasses.field(default_factory=dict)
    submodules: object = None


def build_module(spec_or_module, *args, **kwargs):
    """megatron.core.transformer.spec_utils.build_module for class-valued specs."""
    if isinstance(spec_or_module, ModuleSpec):
        if spec_or_module.submodules is not None:
            kwargs["submodules"] = spec_or_module.submodules
        return spec_or_module.module(*args, **spec_or_module.params, **kwargs)
    return spec_or_module(*args, **kwargs)


def make_viewless_tensor(inp, requires_grad, keep_graph):
    """megatron.core.utils.make_viewless_tensor: same values, same graph (the original only drops the `._base` reference)."""
    return inp


class Layer(torch.nn.Module, BaseTransformerLayer):
    """A transformer layer as TransformerBlock builds (config=, layer_number=) and calls it -> (hidden_states, context)."""

    def __init__(self, config, layer_number=1, **kw):
        super().__init__()
        self.layer_number = layer_number
        self.lin = torch.nn.Linear(config.hidden_size, config.hidden_size)
        self.calls = 0

    def forward(self, hidden_states, attention_mask=None, context=None, context_mask=None, rotary_pos_emb=None, inference_params=None,
                packed_seq_params=None):
        self.calls += 1
        y = torch.tanh(self.lin(hidden_states)) * (0.5 + 0.1 * self.layer_number) + hidden_states
        if rotary_pos_emb is not None:
            y = y + 0.01 * rotary_pos_emb[:y.size(0), :, 0, :1]
        return y, context


class Norm(torch.nn.LayerNorm):
    """TENorm's constructor: (config=, hidden_size=, eps=)."""

    def __init__(self, config=None, hidden_size=None, eps=1e-5):
        super().__init__(hidden_size, eps=eps)
