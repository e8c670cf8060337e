"""The rest of the vision path as Megatron-constructible, HIP-backed modules — what `pretrain_long_vita.py` /
`run_text_generation_server.py` reach when both adaptors are imported (VERDICT r3 "missing" 1):

  InternViTModel / SigLIPViTModel   M/core/models/vision/intern_vit_model.py:91-261, siglip_vit_model.py:89-228 — the reference's
        constructor signature, Parameters under its names (`conv1.weight / .bias`, `position_embeddings.weight`, `class_token`,
        `decoder.*`), `forward(x, attention_mask) -> [b, s, h]`.  The front end (Conv2d 14 / 14 -> reshape / permute -> cat(class
        token) -> + position embedding -> permute to [s, b, h]) is vita_patchify14 -> vita_gemm_bf16(+ bias) -> vita_vit_assemble
        writing Megatron's [s, b, h] directly, with its backward (stage 2 trains the encoder): d conv weight / bias through the
        TN GEMM, d position table / d class token through vita_vit_assemble_bwd.  `decoder` is Megatron's own TransformerBlock
        built from the (HIP) layer spec.
  MultimodalProjector               M/core/models/vision/multimodal_projector.py:12-80 — `encoder.linear_fc1 / linear_fc2` (no bias),
        GELU between them: layers.ViTMLP over HIP linears instead of Megatron's MLP (whose RowParallelLinear is torch.matmul).
  install_hip_vision_path(model)    MegatronVisionModel's forward_once / forward_downsample / forward_projection
        (M/pretrain_long_vita.py:436-520) rebound on the instance: drop class token + pixel_shuffle(0.5) + LayerNorm(4 h) as ONE
        kernel reading the encoder's [s, b, h] output in place (vita_pixel_shuffle_ln_ex), with its backward
        (vita_pixel_shuffle_ln_bwd: dx scattered back through the shuffle, d gamma / d beta), then the projector.  The class lives
        in the reference's ENTRY SCRIPT (`__main__` under training), so it is reached through the object GPTVLModel builds:
        `gpt_vl_model_init_wrapper` (registered on GPTVLModel.__init__) calls this on `self.external_feature_model`.
        `pre_proj_layernorm` stays the torch.nn.LayerNorm the reference constructs — as a parameter container only.
  vocab_parallel_cross_entropy      megatron.core.tensor_parallel.cross_entropy (called through
        LanguageModule.compute_language_model_loss at gpt_vl_model.py:414): TP = 1 vita_ce_loss(_f32) forward, the same row pass with
        grad_scale = incoming gradient backward; TP > 1 the vocabulary-parallel form on the local shard (vita_ce_vp_stats -> one
        all-gather of 16-byte row records -> vita_ce_vp_finish; vita_ce_vp_grad).  Targets outside the vocabulary (the -100 padding)
        are masked as Megatron masks them.

`final_layernorm` needs no class of its own: megatron_adaptor registers layers.Norm on
`megatron.core.transformer.custom_layers.transformer_engine.TENorm`, the name TransformerBlock builds it from
(M/core/transformer/transformer_block.py:201; the reference's adaptor has the same patch, commented out, with its PTNorm)."""
from __future__ import annotations

import types
from contextlib import nullcontext
from functools import wraps
from typing import Optional

import torch
import torch.distributed as dist
from torch.nn import Parameter

from . import autograd_fns as F_, ops, parallel_state as mpu
from .layers import ColumnParallelLinear, RowParallelLinear, ViTMLP


# ------------------------------------------------------------------------------------------------------------------------------
# patch embedding + class token + position embedding  ->  [s, b, h]
# ------------------------------------------------------------------------------------------------------------------------------
def _padded_conv_weight(conv_w: torch.Tensor, k_pad: int) -> torch.Tensor:
    """Conv2d.weight [h, 3, 14, 14] -> [h, k_pad] (columns 588.. zero): the GEMM steps its contraction in 64s."""
    h = conv_w.shape[0]
    w = torch.zeros(h, k_pad, dtype=conv_w.dtype, device=conv_w.device)
    w[:, :588] = conv_w.detach().reshape(h, 588)
    return w


class PatchEmbedFn(torch.autograd.Function):
    """images [n, 3, H, W] bf16, conv weight [h, 3, 14, 14], conv bias [h], class token [1, 1, h] | None, position table [rows, h]
    -> x [seq, n, h] = Megatron's [s, b, h] (intern_vit_model.py:203-216,241-243)."""

    K_FWD, K_TRAIN = 640, 768        # 588 padded to the GEMM's K step; to a multiple of 256 where the TN wgrad kernel needs it

    @staticmethod
    def forward(ctx, images, conv_w, conv_b, cls, pos_table, pos_row0):
        n, _, H, W = images.shape
        n_patches = (H // 14) * (W // 14)
        patches = ops.patchify14(images, PatchEmbedFn.K_FWD, token_major=True)
        pe = ops.gemm(patches, _padded_conv_weight(conv_w, PatchEmbedFn.K_FWD), ops.EPI_BIAS if conv_b is not None else ops.EPI_NONE, conv_b)
        del patches
        x = ops.vit_assemble(pe, None if cls is None else cls.reshape(-1), pos_table, n, n_patches, pos_row0, token_major=True)
        ctx.save_for_backward(images)                # the patches are re-derived in the backward (0.6 GB per 256 frames otherwise)
        ctx.meta = (conv_w.shape, conv_w.dtype, conv_b is not None, cls is not None, tuple(pos_table.shape), pos_row0)
        return x

    @staticmethod
    def backward(ctx, g):
        (images,) = ctx.saved_tensors
        w_shape, w_dtype, has_bias, has_cls, pos_shape, pos_row0 = ctx.meta
        g = g.contiguous()
        seq, n, h = g.shape
        d_w = d_b = d_cls = d_pos = None
        if ctx.needs_input_grad[3] or ctx.needs_input_grad[4]:
            rows = ops.vit_assemble_bwd(g, token_major=True)                     # [seq, h]: sum over the images
            if ctx.needs_input_grad[4]:
                d_pos = torch.zeros(pos_shape, dtype=g.dtype, device=g.device)
                d_pos[pos_row0:pos_row0 + seq] = rows
            if has_cls and ctx.needs_input_grad[3]:
                d_cls = rows[0].reshape(1, 1, h).clone()
        if ctx.needs_input_grad[1] or (has_bias and ctx.needs_input_grad[2]):
            d_pe = (g[1:] if has_cls else g).reshape(-1, h)                      # token-major rows, like the patches below: no transposes
            patches = ops.patchify14(images, PatchEmbedFn.K_TRAIN, token_major=True)
            d_wp, d_b = F_.weight_bias_grads(d_pe, patches, ctx.needs_input_grad[1], has_bias and ctx.needs_input_grad[2], h, w_dtype)
            if d_wp is not None:
                d_w = d_wp[:, :588].reshape(w_shape)
        return None, d_w, d_b, d_cls, d_pos, None


class _HipViTModel(torch.nn.Module):
    """Shared body of InternViTModel / SigLIPViTModel (the two reference classes differ in the class token and the position ids)."""

    def __init__(self, transformer_config, transformer_layer_spec, add_class_token, class_token_len, patch_dim, img_h, img_w,
                 vision_context_parallel, siglip: bool):
        super().__init__()
        from megatron.core.transformer.transformer_block import TransformerBlock
        if patch_dim != 14:
            raise NotImplementedError("the patch-embedding kernel is built for 14 x 14 patches (every Long-VITA vision tower)")
        if add_class_token and class_token_len != 1:
            raise NotImplementedError("class_token_len != 1")
        self.config = transformer_config
        self.class_token_len = class_token_len
        self.visual_hidden_size = transformer_config.hidden_size
        self.patch_dim, self.img_h, self.img_w = patch_dim, img_h, img_w
        if not siglip:
            assert self.img_h % self.patch_dim == 0
            assert self.img_w % self.patch_dim == 0
        self.num_patches_per_dim_h = self.img_h // self.patch_dim
        self.num_patches_per_dim_w = self.img_w // self.patch_dim
        self.num_patches = self.num_patches_per_dim_h * self.num_patches_per_dim_w
        self.add_class_token = bool(add_class_token) and not siglip
        self.seq_length = self.num_patches + (self.class_token_len if self.add_class_token else 0)
        dtype = getattr(transformer_config, "params_dtype", torch.bfloat16)
        dev = None if getattr(transformer_config, "use_cpu_initialization", False) else torch.cuda.current_device()
        h = self.visual_hidden_size
        self.conv1 = torch.nn.Conv2d(3, h, kernel_size=patch_dim, stride=patch_dim, bias=True, device=dev, dtype=dtype)  # parameters only
        if siglip:                                                   # siglip_vit_model.py:135-137
            self.pos_row0, rows = 0, self.seq_length
        elif self.add_class_token:                                   # intern_vit_model.py:140,144
            self.pos_row0, rows = 0, self.seq_length
        else:                                                        # :141-146: ids 1 .. seq of a table with seq + 1 rows
            self.pos_row0, rows = 1, self.seq_length + 1
        self.position_ids = torch.arange(self.pos_row0, self.pos_row0 + self.seq_length, device=dev).expand(1, -1)
        self.position_embeddings = torch.nn.Embedding(rows, h, device=dev, dtype=dtype)
        if not siglip:
            self.class_token = Parameter(torch.randn(1, self.class_token_len, h, device=dev, dtype=dtype))
            if not self.add_class_token:
                self.class_token.requires_grad = False               # :156-160
        self.decoder = TransformerBlock(config=transformer_config, spec=transformer_layer_spec, pre_process=True, post_process=False)
        self.vision_context_parallel = vision_context_parallel
        self.vita_keep_sbh = False      # install_hip_vision_path: the consumer reads the [s, b, h] memory through strides — no copy

    def set_input_tensor(self, input_tensor: torch.Tensor) -> None:
        self.decoder.set_input_tensor(input_tensor)

    def forward(self, x: torch.Tensor, attention_mask: Optional[torch.Tensor] = None) -> torch.Tensor:
        """x [batch, 3, img_h, img_w] -> [b, s, h] (intern_vit_model.py:190-261 / siglip_vit_model.py:165-228)."""
        if self.vision_context_parallel and mpu.get_context_parallel_world_size() != 1:
            raise NotImplementedError("--vision-context-parallel (no reference script sets it) is not built")
        if x.dtype != torch.bfloat16:
            x = x.to(torch.bfloat16)                                 # get_batch_on_this_tp_rank hands bf16 (M/training/utils.py:450)
        assert (x.shape[2] // 14) * (x.shape[3] // 14) == self.num_patches, f"{x.shape} != {self.seq_length}"
        cls = self.class_token if self.add_class_token else None
        args = (x, self.conv1.weight, self.conv1.bias, cls, self.position_embeddings.weight, self.pos_row0)
        if torch.is_grad_enabled() and any(t is not None and t.requires_grad for t in args[1:5]):
            h = PatchEmbedFn.apply(*args)
        else:
            h = PatchEmbedFn.forward(_NoCtx(), *args)
        h = self.decoder(h, attention_mask)                          # [s, b, h]
        out = h.permute(1, 0, 2)                                     # [s, b, h] -> [b, s, h]
        return out if self.vita_keep_sbh else out.contiguous()


class _NoCtx:
    """Inference call of an autograd Function's forward: nothing is saved."""
    needs_input_grad = ()

    def save_for_backward(self, *a):
        pass


class InternViTModel(_HipViTModel):
    def __init__(self, transformer_config, transformer_layer_spec, add_class_token: bool = True, class_token_len: int = 1,
                 patch_dim: int = 14, img_h: int = 336, img_w: int = 336, vision_context_parallel: bool = False) -> None:
        super().__init__(transformer_config, transformer_layer_spec, add_class_token, class_token_len, patch_dim, img_h, img_w,
                         vision_context_parallel, siglip=False)


class SigLIPViTModel(_HipViTModel):
    def __init__(self, transformer_config, transformer_layer_spec, add_class_token: bool = None, class_token_len: int = None,
                 patch_dim: int = 14, img_h: int = 384, img_w: int = 384, vision_context_parallel: bool = False) -> None:
        super().__init__(transformer_config, transformer_layer_spec, False, 1, patch_dim, img_h, img_w, vision_context_parallel,
                         siglip=True)


# ------------------------------------------------------------------------------------------------------------------------------
# projector
# ------------------------------------------------------------------------------------------------------------------------------
class MultimodalProjector(torch.nn.Module):
    """`MultimodalProjector(config, submodules, projector_type, input_size)`; forward(hidden_states) -> hidden_states
    (multimodal_projector.py:27-80).  `submodules` (Megatron's MLPSubmodules) names the linears' classes; whatever it names, the
    leaves built here are this package's ColumnParallelLinear / RowParallelLinear so that both GEMMs and the GELU run on the library."""

    def __init__(self, config, submodules, projector_type: str, input_size: int):
        super().__init__()
        self.config, self.projector_type = config, projector_type
        assert submodules is not None, "MLPSubmodules must be provided"
        hip = types.SimpleNamespace(linear_fc1=ColumnParallelLinear, linear_fc2=RowParallelLinear)
        if projector_type == "mlp":
            self.encoder = ViTMLP(config=config, submodules=hip, input_size=input_size)
            self.encoder.layer_number = 1
        elif projector_type == "affine":
            self.encoder = ColumnParallelLinear(input_size, config.hidden_size, config=config, init_method=config.init_method,
                                                gather_output=True, bias=config.add_bias_linear, skip_bias_add=True, is_expert=False,
                                                tp_comm_buffer_name=None)
        else:
            raise Exception(f"Unsupported multimodal projection type {self.projector_type}")

    def forward(self, hidden_states):
        encoder_output, encoder_output_bias = self.encoder(hidden_states)
        if encoder_output_bias is not None:
            from .layers import BiasAddFn
            encoder_output = BiasAddFn.apply(encoder_output, encoder_output_bias)
        return encoder_output


# ------------------------------------------------------------------------------------------------------------------------------
# forward_downsample + forward_projection of MegatronVisionModel
# ------------------------------------------------------------------------------------------------------------------------------
class PixelShuffleLNFn(torch.autograd.Function):
    """x [n, seq, h] (any image / token stride) -> [n, (grid/2)^2, 4 h]: vit_output[:, 1:] -> pixel_shuffle(0.5) -> LayerNorm(4 h)
    (M/pretrain_long_vita.py:452-461,572-582,436-441); weight None: the permutation alone."""

    @staticmethod
    def forward(ctx, x, weight, bias, grid, has_cls, eps):
        ctx.save_for_backward(x, weight)
        ctx.meta = (grid, has_cls, eps, bias is not None)
        return ops.pixel_shuffle_ln(x, weight, bias, grid, has_cls, eps, norm=weight is not None)

    @staticmethod
    def backward(ctx, dy):
        x, weight = ctx.saved_tensors
        grid, has_cls, eps, has_bias = ctx.meta
        norm = weight is not None
        dg = db = None
        if norm:
            dg = torch.zeros(weight.numel(), dtype=torch.float32, device=x.device)
            db = torch.zeros_like(dg)
        dx = ops.pixel_shuffle_ln_bwd(dy, x, weight, grid, has_cls, eps, dg, db, want_dx=ctx.needs_input_grad[0], norm=norm)
        return (dx, None if dg is None else dg.to(weight.dtype), (db.to(weight.dtype) if (norm and has_bias) else None), None, None, None)


def _grid_of(model, vit_output) -> int:
    seq = vit_output.shape[1] - (1 if model.add_class_token else 0)
    g = int(round(seq ** 0.5))
    if g * g != seq or g % 2:
        raise ValueError(f"vision sequence {vit_output.shape[1]} is not an even square grid (+ class token)")
    return g


def _fusable(model) -> bool:
    """The configuration every reference script runs: --vision-downsample-ratio 0.5, stride 1 (stage3 .sh:205)."""
    return model.vision_downsample_ratio == 0.5 and model.vision_downsample_stride == 1 and model.add_class_token in (True, False)


def hip_forward_downsample(self, vit_output):
    """forward_downsample (M/pretrain_long_vita.py:452-470): drop the class token + pixel_shuffle(0.5) as one permutation kernel."""
    if not _fusable(self) or vit_output.shape[2] * 4 > 4096:
        # the reference's pixel_shuffle `view`s its input (M/pretrain_long_vita.py:452-470): the encoder's strided [b, s, h] view of
        # [s, b, h] memory (vita_keep_sbh) has to be made contiguous first (hidden * 4 > 4096, e.g. InternViT-6B)
        return self._vita_ref_forward_downsample(vit_output.contiguous())
    if torch.is_grad_enabled() and vit_output.requires_grad:
        return PixelShuffleLNFn.apply(vit_output, None, None, _grid_of(self, vit_output), bool(self.add_class_token), 0.0)
    return ops.pixel_shuffle_ln(vit_output, None, None, _grid_of(self, vit_output), bool(self.add_class_token), 0.0, norm=False)


def _pre_norm(self, x):
    ln = self.pre_proj_layernorm
    if isinstance(ln, torch.nn.Identity):
        return x
    if torch.is_grad_enabled() and (x.requires_grad or ln.weight.requires_grad):
        return F_.LayerNormFn.apply(x, ln.weight, ln.bias, ln.eps)
    return ops.layernorm(x, ln.weight, ln.bias, ln.eps)


def hip_forward_projection(self, vit_output):
    """forward_projection (:436-450): pre_proj_layernorm (vita_layernorm_fwd / _bwd on the torch module's parameters) + projector."""
    return self.vision_projection(_pre_norm(self, vit_output))


def _downsample_project(self, vit_output):
    """Both steps with the shuffle and the norm in ONE kernel: vit_output [b, s, h] (a view of the encoder's [s, b, h] memory)."""
    ln = self.pre_proj_layernorm
    grid = _grid_of(self, vit_output)
    if torch.is_grad_enabled() and (vit_output.requires_grad or ln.weight.requires_grad):
        t = PixelShuffleLNFn.apply(vit_output, ln.weight, ln.bias, grid, bool(self.add_class_token), ln.eps)
    else:
        t = ops.pixel_shuffle_ln(vit_output, ln.weight, ln.bias, grid, bool(self.add_class_token), ln.eps)
    return self.vision_projection(t)


def hip_forward_once(self, images, attention_mask):
    """forward_once (:485-520) — the freeze contexts and the two recompute switches of the reference; with the reference's flags
    (ratio 0.5, `--vision-projector-pre-norm`) downsample + norm fuse into one kernel in front of the projector."""
    fused = _fusable(self) and isinstance(self.pre_proj_layernorm, torch.nn.LayerNorm) and self.pre_proj_layernorm.weight.numel() <= 4096
    with (torch.no_grad() if self.vision_model_freeze else nullcontext()):
        vit_output = self.vit(images, attention_mask)
        if not fused:
            if self.vision_model_recompute:
                from megatron.core import tensor_parallel
                vit_output = tensor_parallel.checkpoint(self.forward_downsample, False, vit_output)
            else:
                vit_output = self.forward_downsample(vit_output)
    fn = (lambda t: _downsample_project(self, t)) if fused else self.forward_projection
    with (torch.no_grad() if self.vision_projector_freeze else nullcontext()):
        if self.vision_projector_recompute:
            from megatron.core import tensor_parallel
            return tensor_parallel.checkpoint(fn, False, vit_output)
        return fn(vit_output)


def install_hip_vision_path(model) -> bool:
    """Rebind the three methods on a MegatronVisionModel INSTANCE (duck-typed: the class is defined in the entry script).  Returns
    False (and touches nothing) for any other external feature model."""
    need = ("forward_once", "forward_downsample", "forward_projection", "pre_proj_layernorm", "vision_projection", "vit",
            "vision_downsample_ratio", "vision_downsample_stride", "add_class_token", "vision_model_freeze", "vision_projector_freeze")
    if model is None or not all(hasattr(model, a) for a in need):
        return False
    if getattr(model, "_vita_hip_installed", False):
        return True
    model._vita_ref_forward_downsample = model.forward_downsample
    model.forward_downsample = types.MethodType(hip_forward_downsample, model)
    model.forward_projection = types.MethodType(hip_forward_projection, model)
    model.forward_once = types.MethodType(hip_forward_once, model)
    if isinstance(model.vit, _HipViTModel):
        model.vit.vita_keep_sbh = _fusable(model)            # the fused kernel reads the [s, b, h] memory through strides
    model._vita_hip_installed = True
    return True


def gpt_vl_model_init_wrapper(fn):
    """Wrapper patch for `long_vita_megatron.core.models.multimodal.gpt_vl_model.GPTVLModel.__init__` (:73-172): once the reference
    has built `self.external_feature_model = external_feature_model_provider(config)` (:113), put its downsample / projection on the
    library.  Both entry points construct the model through this class (pretrain_long_vita.py:640-655, run_text_generation_server.py:66-82)."""
    @wraps(fn)
    def wrapper(self, *args, **kwargs):
        fn(self, *args, **kwargs)
        install_hip_vision_path(getattr(self, "external_feature_model", None))

    return wrapper


# ------------------------------------------------------------------------------------------------------------------------------
# vocabulary-parallel cross entropy
# ------------------------------------------------------------------------------------------------------------------------------
class VocabParallelCrossEntropyFn(torch.autograd.Function):
    """logits [s, b, V / TP] (fp32 as Megatron passes them, or bf16), target [s, b] int64 in global vocabulary ids -> loss [s, b] fp32.
    megatron/core/tensor_parallel/cross_entropy.py in its vocabulary-parallel form: every rank keeps only its [rows, V / TP] shard;
    one row pass produces {max, sum-exp, predicted logit, holds-the-label} per row, the TP group all-gathers those 16-byte records (ONE
    small collective where Megatron runs three all-reduces), and the backward writes the local slice of the gradient only.  Targets
    outside the vocabulary (IGNORE_TOKEN_ID = -100 padding, M/pretrain_long_vita.py:751) are masked as Megatron masks them —
    predicted logit 0, no one-hot term — and left to the caller's loss mask; nothing raises and nothing synchronises."""

    @staticmethod
    def forward(ctx, vocab_parallel_logits, target, label_smoothing=0.0):
        if label_smoothing:
            raise NotImplementedError("label smoothing is not on the Long-VITA path")
        tp, group = mpu.get_tensor_model_parallel_world_size(), mpu.get_tensor_model_parallel_group()
        shape = target.shape
        v_l = vocab_parallel_logits.shape[-1]
        local = vocab_parallel_logits.reshape(-1, v_l)
        labels = target.reshape(-1).contiguous()
        if tp == 1:
            loss = ops.ce_loss(local, labels, strict=False)
            ctx.save_for_backward(local, labels)
            ctx.meta = (1, 0, tuple(vocab_parallel_logits.shape))
            return loss.view(shape)
        start = mpu.get_tensor_model_parallel_rank() * v_l
        stats = ops.ce_vp_stats(local, labels, start)
        stats_all = torch.empty((tp,) + tuple(stats.shape), dtype=stats.dtype, device=stats.device)
        dist.all_gather_into_tensor(stats_all.view(-1), stats.view(-1), group=group)
        loss, row_stat = ops.ce_vp_finish(stats_all)
        ctx.save_for_backward(local, labels, row_stat)
        ctx.meta = (tp, start, tuple(vocab_parallel_logits.shape))
        return loss.view(shape)

    @staticmethod
    def backward(ctx, g):
        tp, start, shape = ctx.meta
        gs = g.reshape(-1).float().contiguous()
        if tp == 1:
            local, labels = ctx.saved_tensors
            _, dl = ops.ce_loss(local, labels, gs, want_grad=True, want_loss=False, strict=False)
        else:
            local, labels, row_stat = ctx.saved_tensors
            dl = ops.ce_vp_grad(local, labels, start, row_stat, gs)
        return dl.reshape(shape), None, None


def vocab_parallel_cross_entropy(vocab_parallel_logits, target, label_smoothing=0.0):
    """megatron.core.tensor_parallel.cross_entropy.vocab_parallel_cross_entropy's signature."""
    return VocabParallelCrossEntropyFn.apply(vocab_parallel_logits, target, label_smoothing)
