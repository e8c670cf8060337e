"""Fixtures from the reference's COMPOSITE classes run on CPU over plain-torch leaves (oracle/leaves.py).  TEST INFRASTRUCTURE ONLY;
runs in the build container only (reads /root/reference), through `python -m oracle.make_golden gptvl_forward vision_model
transformer_block intern_vit_forward forward_step`.

    gptvl_forward.pt       M/core/models/multimodal/gpt_vl_model.py:72-416  GPTVLModel.__init__ / forward / the three freeze methods (imported)
    transformer_block.pt   M/core/transformer/transformer_block.py:119-420  TransformerBlock (imported): layer loop, full recompute
                           ("block" / "uniform"), final norm, pre/post_process
    intern_vit_forward.pt  M/core/models/vision/intern_vit_model.py:92-261  InternViTModel.__init__ / forward (imported): conv1, class
                           token, position embeddings, [b s h] <-> [s b h]
    vision_model.pt        M/pretrain_long_vita.py:310-596  MegatronVisionModel (class source executed): constructor wiring, downsample,
                           projection, freeze / recompute contexts, 256-image chunks
    forward_step.pt        M/pretrain_long_vita.py:841-869  forward_step (function source executed): --logit-mask handling

Each case stores its seeds / flags, the outputs, and the gradient of every parameter (None where autograd reaches nothing), so that
`tests/test_oracle_golden.py` can require `tests/dummy_megatron.py`'s hand restatements — the classes the GPU boundary tests compose the
product's modules with — to reproduce the reference's own code bit for bit in fp32."""
from __future__ import annotations

import ast
import contextlib
import importlib
import os
import sys
import types
from functools import partial
from unittest import mock

import torch

from . import leaves

REF = "/root/reference"
M = os.path.join(REF, "long_vita_megatron")


def _fresh_import(name: str, bindings: dict):
    """Import (again) a reference module after binding leaf objects on the stub `megatron.*` modules it imports from.
    bindings: {"megatron.core.x.y": {"Name": obj}}."""
    for mod, names in bindings.items():
        importlib.import_module(mod)                                   # the stub finder fabricates it
        for k, v in names.items():
            setattr(sys.modules[mod], k, v)
    sys.modules.pop(name, None)
    return importlib.import_module(name)


def compact(t):
    """Fixtures stay small: a tensor of more than 4096 elements is stored as its shape + the SHA-256 of its bytes (the comparisons are
    bit-exact, so a digest loses nothing) + its first 32 values for a readable failure."""
    if t is None or not torch.is_tensor(t) or t.numel() <= 4096:
        return t if t is None or not torch.is_tensor(t) else t.detach().clone()
    import hashlib
    c = t.detach().contiguous()
    return {"shape": tuple(c.shape), "dtype": str(c.dtype), "sha256": hashlib.sha256(c.numpy().tobytes()).hexdigest(),
            "head": c.reshape(-1)[:32].clone()}


def same(got, want) -> bool:
    """Bit-exact comparison of a tensor with a stored tensor or digest (None == None)."""
    if want is None or got is None:
        return want is None and got is None
    if isinstance(want, dict):
        return compact(got)["sha256"] == want["sha256"] and tuple(got.shape) == tuple(want["shape"])
    return got.shape == want.shape and torch.equal(got, want)


def _grads(module):
    return {n: (None if p.grad is None else compact(p.grad)) for n, p in module.named_parameters()}


def _source_of(path: str, *names):
    """The source segments of top-level definitions of a reference file that cannot be imported whole."""
    src = open(path).read()
    body = {n.name: n for n in ast.parse(src).body if isinstance(n, (ast.FunctionDef, ast.ClassDef))}
    return "\n\n".join(ast.get_source_segment(src, body[n]) for n in names)


# ------------------------------------------------------------------------------------------------------------------------------
# GPTVLModel
# ------------------------------------------------------------------------------------------------------------------------------
GPTVL_CASES = [
    dict(name="train_mask_instruction", labels=True, logit_mask=True, instruction=True),
    dict(name="train_mask_plain", labels=True, logit_mask=True, instruction=False),
    dict(name="train_nomask_instruction", labels=True, logit_mask=False, instruction=True),
    dict(name="train_softcap_scale", labels=True, logit_mask=True, instruction=True, softcap=3.0, scale=0.5),
    dict(name="train_images", labels=True, logit_mask=True, instruction=True, images=True),
    dict(name="train_images_pre_len", labels=True, logit_mask=False, instruction=False, images=True, pre_len=3),
    dict(name="train_learned_absolute", labels=True, logit_mask=True, instruction=True, rope=False),
    dict(name="infer_logits", labels=False, logit_mask=False),
    dict(name="infer_logits_mask", labels=False, logit_mask=True, softcap=2.0),
    dict(name="infer_params_override", labels=False, logit_mask=False, images=True, inference_params=dict(use_kv_cache=False)),
    dict(name="infer_params_kv_cache", labels=False, logit_mask=False, images=True, inference_params=dict(use_kv_cache=True)),
    dict(name="infer_params_filled_cache", labels=False, logit_mask=False, inference_params=dict(use_kv_cache=True, filled=True)),
]
GPTVL_SIZES = dict(seq=12, vocab=40, hidden=16)


def gptvl_inputs(case: dict):
    """Seeded inputs of one case: tokens / labels / masks / the external inputs (images + indices) / the inference-params object."""
    S, V = GPTVL_SIZES["seq"], GPTVL_SIZES["vocab"]
    g = torch.Generator().manual_seed(100 + len(case["name"]))
    tokens = torch.randint(0, V, (1, S), generator=g)
    labels = torch.randint(0, V, (1, S), generator=g)
    position_ids = torch.arange(S).unsqueeze(0)
    mask = torch.zeros(1, S, dtype=torch.bool)
    mask[0, [2, 3, 7, 8, 9, 11]] = True
    ext = {}
    if case.get("images"):
        ext = {"images": torch.randn(2, 3, 2, 4, generator=g), "image_indices": torch.tensor([1, 4, 5, 6]), "other": torch.zeros(1)}
        if "pre_len" in case:
            ext["pre_len"] = case["pre_len"]
    ip = None
    if case.get("inference_params") is not None:
        spec = case["inference_params"]
        ip = types.SimpleNamespace(external_inputs=ext or None, key_value_memory_dict={0: "k"} if spec.get("filled") else {},
                                   logit_mask=mask.clone(), use_kv_cache=spec["use_kv_cache"])
    return tokens, labels, position_ids, mask, ext, ip


class GptvlEmbedding(leaves.Embedding):
    """The leaf embedding reading the key the case uses for its indices."""

    def forward(self, input_ids, position_ids, external_feature_dict=None):
        if external_feature_dict:
            external_feature_dict = dict(external_feature_dict, indices=external_feature_dict["image_indices"])
        return super().forward(input_ids, position_ids, external_feature_dict)


def run_gptvl_case(cls, case: dict, args_ns) -> dict:
    """Build `cls` (the reference's GPTVLModel or a restatement with its constructor) over the leaves, run one case."""
    cfg = leaves.config(hidden_size=GPTVL_SIZES["hidden"])
    args_ns.output_multiplier_scale = case.get("scale")
    args_ns.output_logit_softcapping = case.get("softcap")
    args_ns.is_instruction_dataset = bool(case.get("instruction"))
    model = cls(config=cfg, transformer_layer_spec="spec", vocab_size=GPTVL_SIZES["vocab"], max_sequence_length=64, pre_process=True,
                post_process=True, parallel_output=True, share_embeddings_and_output_weights=False,
                position_embedding_type="rope" if case.get("rope", True) else "learned_absolute", rotary_percent=1.0, rotary_base=1000,
                external_feature_model_provider=leaves.FeatureModel, external_args=("a", 2))
    leaves.init_by_name(model, seed=7)
    model.train(bool(case["labels"]))
    tokens, labels, position_ids, mask, ext, ip = gptvl_inputs(case)
    kw = {}
    if ip is not None:
        kw["inference_params"] = ip
    out = model(tokens, position_ids, None, labels=labels if case["labels"] else None,
                external_inputs={} if ip is not None else ext, logit_mask=mask if case["logit_mask"] else None, **kw)
    res = {"out": out.detach().clone(), "external_args": model.external_feature_model.external_args,
           "output_layer_built_with": model.output_layer.built_with, "decoder_saw_inference_params": model.decoder.seen["inference_params"] is not None,
           "param_names": [n for n, _ in model.named_parameters()]}
    if out.requires_grad:
        w = torch.linspace(0.5, 1.5, out.numel()).view_as(out)
        (out * w).sum().backward()
        res["grads"] = _grads(model)
    return res


def freeze_report(cls) -> dict:
    out = {}
    for method in ("vision_projector_freeze", "vision_model_freeze", "language_model_freeze"):
        model = cls(config=leaves.config(hidden_size=GPTVL_SIZES["hidden"]), transformer_layer_spec="spec", vocab_size=GPTVL_SIZES["vocab"],
                    max_sequence_length=64, position_embedding_type="rope", external_feature_model_provider=leaves.FeatureModel)
        with contextlib.redirect_stdout(open(os.devnull, "w")):
            ret = getattr(model, method)()
        assert ret is model
        out[method] = sorted(n for n, p in model.named_parameters() if not p.requires_grad)
    return out


def golden_gptvl_forward(out_dir: str, state: dict):
    mod = _fresh_import("long_vita_megatron.core.models.multimodal.gpt_vl_model", {
        "megatron.core.models.common.language_module.language_module": {"LanguageModule": leaves.LanguageModule},
        "megatron.core.models.common.embeddings.language_model_embedding": {"LanguageModelEmbedding": GptvlEmbedding},
        "megatron.core.models.common.embeddings.rotary_pos_embedding": {"RotaryEmbedding": leaves.Rotary},
        "megatron.core.transformer.transformer_block": {"TransformerBlock": leaves.Block},
        "megatron.core.tensor_parallel": {"ColumnParallelLinear": leaves.ColumnParallelLinear},
    })
    args = state["args"]
    res = {"cases": []}
    with mock.patch.object(torch.distributed, "get_rank", lambda *a, **k: 0), contextlib.redirect_stdout(open(os.devnull, "w")):
        for case in GPTVL_CASES:
            res["cases"].append(dict(case, **run_gptvl_case(mod.GPTVLModel, case, args)))
    res["freeze"] = freeze_report(mod.GPTVLModel)
    torch.save(res, os.path.join(out_dir, "gptvl_forward.pt"))


# ------------------------------------------------------------------------------------------------------------------------------
# TransformerBlock
# ------------------------------------------------------------------------------------------------------------------------------
BLOCK_CASES = [
    dict(name="plain", num_layers=3),
    dict(name="no_final_norm", num_layers=2, post_process=False),
    dict(name="rotary", num_layers=2, rotary=True),
    dict(name="recompute_block_1_of_3", num_layers=3, granularity="full", method="block", n=1, rotary=True),
    dict(name="recompute_block_all", num_layers=3, granularity="full", method="block", n=3),
    dict(name="recompute_uniform_2_of_4", num_layers=4, granularity="full", method="uniform", n=2),
    dict(name="recompute_in_eval_is_off", num_layers=3, granularity="full", method="block", n=2, eval=True),
    dict(name="not_pre_process_reads_input_tensor", num_layers=2, pre_process=False),
]


def run_block_case(cls, spec_cls, case: dict) -> dict:
    cfg = leaves.config(hidden_size=16, num_layers=case["num_layers"], recompute_granularity=case.get("granularity"),
                        recompute_method=case.get("method"), recompute_num_layers=case.get("n"))
    blk = cls(cfg, spec_cls(module=leaves.Layer), pre_process=case.get("pre_process", True), post_process=case.get("post_process", True))
    leaves.init_by_name(blk, seed=3)
    blk.train(not case.get("eval", False))
    g = torch.Generator().manual_seed(5)
    x = torch.randn(6, 2, 16, generator=g).requires_grad_(True)
    rot = leaves.Rotary(8, rotary_base=50)(6) if case.get("rotary") else None
    if not case.get("pre_process", True):
        blk.set_input_tensor(x)
        out = blk(torch.full_like(x, 99.0), None, rotary_pos_emb=rot)
    else:
        out = blk(x, None, rotary_pos_emb=rot)
    w = torch.linspace(-1, 1, out.numel()).view_as(out)
    (out * w).sum().backward()
    return {"out": out.detach().clone(), "dx": x.grad.clone(), "grads": _grads(blk), "layer_calls": [l.calls for l in blk.layers],
            "has_final_norm": hasattr(blk, "final_layernorm"), "param_names": [n for n, _ in blk.named_parameters()]}


def golden_transformer_block(out_dir: str, state: dict):
    mod = _fresh_import("long_vita_megatron.core.transformer.transformer_block", {
        "megatron.core.transformer.custom_layers.transformer_engine": {"TENorm": leaves.Norm, "get_cpu_offload_context": None},
        "megatron.core.transformer.module": {"MegatronModule": leaves.MegatronModule},
        "megatron.core.transformer.spec_utils": {"ModuleSpec": leaves.ModuleSpec, "build_module": leaves.build_module},
        "megatron.core.transformer.transformer_layer": {"BaseTransformerLayer": leaves.BaseTransformerLayer, "TransformerLayer": leaves.Layer},
        "megatron.core.utils": {"make_viewless_tensor": leaves.make_viewless_tensor},
        "megatron.core.tensor_parallel": {"checkpoint": leaves.checkpoint},
        "megatron.core": {"tensor_parallel": sys.modules["megatron.core.tensor_parallel"]},
    })
    res = {"cases": [dict(c, **run_block_case(mod.TransformerBlock, leaves.ModuleSpec, c)) for c in BLOCK_CASES]}
    torch.save(res, os.path.join(out_dir, "transformer_block.pt"))


# ------------------------------------------------------------------------------------------------------------------------------
# InternViTModel.forward
# ------------------------------------------------------------------------------------------------------------------------------
VIT_CASES = [dict(name="class_token", add_class_token=True, img=56, patch=14, hidden=32, frames=3),
             dict(name="no_class_token", add_class_token=False, img=42, patch=14, hidden=32, frames=2)]


def vit_case_inputs(case):
    g = torch.Generator().manual_seed(40 + case["img"])
    return torch.randn(case["frames"], 3, case["img"], case["img"], generator=g)


def golden_intern_vit_forward(out_dir: str, state: dict, cpu_as_cuda):
    mod = _fresh_import("long_vita_megatron.core.models.vision.intern_vit_model", {
        "megatron.core.models.common.vision_module.vision_module": {"VisionModule": leaves.MegatronModule},
        "megatron.core.transformer.transformer_block": {"TransformerBlock": leaves.Block},
        "megatron.core.transformer.transformer_layer": {"TransformerLayer": leaves.Layer, "TransformerLayerSubmodules": object,
                                                        "make_viewless_tensor": leaves.make_viewless_tensor},
        "megatron.core.transformer.custom_layers.transformer_engine": {"TENorm": leaves.Norm},
    })
    mod.mpu.get_context_parallel_world_size = lambda: 1
    res = {"cases": []}
    for case in VIT_CASES:
        with cpu_as_cuda():
            vit = mod.InternViTModel(leaves.config(hidden_size=case["hidden"]), "spec", add_class_token=case["add_class_token"],
                                     patch_dim=case["patch"], img_h=case["img"], img_w=case["img"])
        leaves.init_by_name(vit, seed=11)
        x = vit_case_inputs(case)
        out = vit(x)
        w = torch.linspace(-1, 1, out.numel()).view_as(out)
        (out * w).sum().backward()
        # the block's input = the front end alone (what the product's patchify -> GEMM -> assemble kernels produce)
        pre = {}
        h = vit.decoder.register_forward_hook(lambda m, a, o: pre.setdefault("x", a[0].detach().clone()))
        with torch.no_grad():
            vit(x)
        h.remove()
        # values, not digests: the GPU test compares the product's bf16 front end with these within a tolerance.  The weights are
        # leaves.init_by_name(seed = 11) over the reference's parameter names (re-derived by the tests, not stored).
        res["cases"].append(dict(case, out=out.detach().clone(), block_input=pre["x"],
                                 grads={n: (None if p.grad is None else p.grad.clone()) for n, p in vit.named_parameters()},
                                 state_keys=sorted(vit.state_dict()), position_ids=vit.position_ids.clone(),
                                 seq_length=vit.seq_length, param_names=[n for n, _ in vit.named_parameters()],
                                 class_token_requires_grad=bool(vit.class_token.requires_grad)))
    torch.save(res, os.path.join(out_dir, "intern_vit_forward.pt"))


# ------------------------------------------------------------------------------------------------------------------------------
# MegatronVisionModel (entry script class) and forward_step
# ------------------------------------------------------------------------------------------------------------------------------
VISION_CASES = [
    dict(name="stage3_trainable", frames=3),
    dict(name="vit_frozen", frames=3, vision_model_freeze=True),
    dict(name="projector_frozen", frames=3, vision_projector_freeze=True),
    dict(name="both_recomputed", frames=3, vision_model_recompute=True, vision_projector_recompute=True),
    dict(name="frozen_vit_recomputed_projector", frames=2, vision_model_freeze=True, vision_projector_recompute=True),
    dict(name="no_pre_norm", frames=2, vision_projector_pre_norm=False),
    dict(name="no_class_token", frames=2, add_class_token=False),
    dict(name="chunks_of_256", frames=259),
    dict(name="siglip", frames=2, vision_model_type="siglip_400m", add_class_token=False),
]


def vision_args(case: dict):
    """The flags of the reference's scripts at toy sizes: image 56 / patch 14 -> grid 4 (+ class token) -> 4 tokens per image."""
    add_cls = case.get("add_class_token", True)
    return types.SimpleNamespace(
        vision_seq_length=16 + (1 if add_cls else 0), image_token_length=4, vision_model_type=case.get("vision_model_type", "intern_300m"),
        vision_context_parallel=False, vision_downsample_ratio=0.5, vision_downsample_stride=1, add_class_token=add_cls,
        vision_model_freeze=case.get("vision_model_freeze", False), vision_projector_freeze=case.get("vision_projector_freeze", False),
        vision_model_recompute=case.get("vision_model_recompute", False), vision_projector_recompute=case.get("vision_projector_recompute", False),
        vision_projector_pre_norm=case.get("vision_projector_pre_norm", True), transformer_impl="transformer_engine", image_size=56,
        hidden_size=24, ffn_hidden_size=48, vit_load=None, patch_dim=14, num_layers=2, num_attention_heads=2, seq_length=64,
        max_position_embeddings=64, add_bias_linear=False, add_qkv_bias=True, swiglu=True, group_query_attention=True, num_query_groups=1,
        normalization="RMSNorm", apply_rope_fusion=True, kv_channels=12, recompute_granularity=None, recompute_method=None,
        recompute_num_layers=None, vision_projector_type="mlp", rank=1)


def run_vision_case(model, case: dict) -> dict:
    leaves.init_by_name(model, seed=13)
    model.train()
    g = torch.Generator().manual_seed(60 + case["frames"])
    images = torch.randn(case["frames"], 3, 56, 56, generator=g)
    out = model(images=images)
    res = {"out": compact(out), "requires_grad": bool(out.requires_grad), "param_names": [n for n, _ in model.named_parameters()],
           "pre_norm_type": type(model.pre_proj_layernorm).__name__, "vit_built_with": {k: v for k, v in model.vit.built_with.items() if k != "spec"},
           "vit_spec": str(model.vit.built_with["spec"]), "projector_built_with": model.vision_projection.built_with}
    if out.requires_grad:
        w = torch.linspace(-1, 1, out.numel()).view_as(out)
        (out * w).sum().backward()
        res["grads"] = _grads(model)
    # the pieces on their own (what the drop-in rebinds): downsample of a [n, s, h] tensor, projection of its result
    with torch.no_grad():
        v = torch.randn(2, model.vision_seq_length, model.vit.built_with["hidden_size"], generator=torch.Generator().manual_seed(61))
        d = model.forward_downsample(v)
        res["downsample_out"], res["projection_out"] = compact(d), compact(model.forward_projection(d))
    return res


def _entry_script_namespace(state: dict):
    """Globals for the executed source of the entry script's definitions: first-party helpers come from the file itself, everything
    Megatron-side is a leaf."""
    path = os.path.join(M, "pretrain_long_vita.py")
    from copy import deepcopy
    from contextlib import nullcontext

    def core_transformer_config_from_args(a, config_class=None):
        return types.SimpleNamespace(**{k: getattr(a, k) for k in ("hidden_size", "ffn_hidden_size", "num_layers", "add_bias_linear")},
                                     gated_linear_unit=a.swiglu, bias_activation_fusion=True, activation_func=torch.nn.functional.silu)

    ns = {"torch": torch, "deepcopy": deepcopy, "nullcontext": nullcontext, "partial": partial, "os": os, "get_args": lambda: state["args"],
          "print_rank_0": lambda *a, **k: None, "print": lambda *a, **k: None, "core_transformer_config_from_args": core_transformer_config_from_args,
          "VisionTransformerConfig": object, "load_checkpoint": None,
          "get_vit_layer_local_spec_for_intern": lambda: "local_spec_for_intern", "get_vit_layer_local_spec_for_siglip": lambda: "local_spec_for_siglip",
          "get_vit_layer_with_transformer_engine_spec_for_intern": lambda: "te_spec_for_intern",
          "get_vit_layer_local_spec_for_eva": lambda: "local_spec_for_eva", "get_vit_layer_spec": lambda use_te=False: "clip_spec",
          "InternViTModel": leaves.ViT, "SigLIPViTModel": leaves.ViT, "EVA2ViTModel": leaves.ViT, "CLIPViTModel": leaves.ViT,
          "get_mlp_module_spec": lambda use_te=False: types.SimpleNamespace(submodules="mlp_submodules"),
          "tensor_parallel": types.SimpleNamespace(checkpoint=leaves.checkpoint),
          "mpu": types.SimpleNamespace(get_context_parallel_world_size=lambda: 1, get_context_parallel_rank=lambda: 0)}
    helpers = ("get_vision_model_args", "get_vision_model_args_intern_300m", "get_vision_model_args_siglip_400m", "MegatronVisionModel",
               "forward_step")
    exec(compile(_source_of(path, *helpers), path, "exec"), ns)
    return ns


def golden_vision_model(out_dir: str, state: dict):
    importlib.import_module("long_vita_megatron.training.utils")                    # print_args (first-party, importable under the stubs)
    sys.modules["long_vita_megatron.core.models.vision.multimodal_projector"] = types.SimpleNamespace(MultimodalProjector=leaves.Projector)
    try:
        ns = _entry_script_namespace(state)
        res = {"cases": []}
        for case in VISION_CASES:
            state["args"] = vision_args(case)
            with contextlib.redirect_stdout(open(os.devnull, "w")):
                model = ns["MegatronVisionModel"](True)
            vit_args = state["args"].vit_args
            r = run_vision_case(model, case)
            r["vit_args"] = {k: getattr(vit_args, k) for k in ("num_layers", "hidden_size", "ffn_hidden_size", "num_attention_heads", "kv_channels",
                                                               "add_bias_linear", "add_qkv_bias", "swiglu", "normalization", "patch_dim",
                                                               "apply_rope_fusion", "group_query_attention") if hasattr(vit_args, k)}
            res["cases"].append(dict(case, **r))
        torch.save(res, os.path.join(out_dir, "vision_model.pt"))
    finally:
        sys.modules.pop("long_vita_megatron.core.models.vision.multimodal_projector", None)


FORWARD_STEP_CASES = [dict(name="logit_mask", logit_mask=True), dict(name="no_logit_mask", logit_mask=False)]


class RecordingModel:
    """What forward_step hands the model, and a [b, n] per-token loss back (n = selected rows - 1 under --logit-mask)."""

    def __init__(self):
        self.calls = []

    def __call__(self, tokens, position_ids, attention_mask, labels=None, external_inputs=None, logit_mask=None):
        self.calls.append(dict(tokens=tokens, position_ids=position_ids, attention_mask=attention_mask, labels=labels,
                               external_inputs=external_inputs, logit_mask=None if logit_mask is None else logit_mask.clone()))
        n = tokens.shape[1] if logit_mask is None else int(logit_mask.sum()) - 1
        return torch.arange(n, dtype=torch.float32).view(1, n) * 0.25 + 1.0


def forward_step_batch():
    g = torch.Generator().manual_seed(77)
    tokens = torch.randint(0, 50, (1, 10), generator=g)
    labels = torch.randint(0, 50, (1, 10), generator=g)
    loss_mask = torch.tensor([[0, 0, 1, 1, 0, 1, 1, 1, 0, 1]], dtype=torch.float32)
    return tokens, labels, loss_mask, None, torch.arange(10).unsqueeze(0), {"images": torch.zeros(1, 3, 2, 2)}


def golden_forward_step(out_dir: str, state: dict):
    ns = _entry_script_namespace(state)
    timer = types.SimpleNamespace(start=lambda: None, stop=lambda: None)

    class STimer:
        def __call__(self, bdata=False):
            return self

        def __enter__(self):
            return self

        def __exit__(self, *a):
            return False

    ns.update(get_timers=lambda: (lambda *a, **k: timer), stimer=STimer(), get_batch=lambda it: forward_step_batch(),
              loss_func=lambda loss_mask, output_tensor: ("loss_func called with", loss_mask, output_tensor))
    res = {"cases": []}
    for case in FORWARD_STEP_CASES:
        state["args"] = types.SimpleNamespace(logit_mask=case["logit_mask"])
        model = RecordingModel()
        out, fn = ns["forward_step"](None, model)
        tag, lm, ot = fn(out)
        call = model.calls[0]
        res["cases"].append(dict(case, out=out, loss_mask_given_to_loss_func=lm, logit_mask_given_to_model=call["logit_mask"],
                                 labels_given=call["labels"], external_keys=sorted(call["external_inputs"])))
    torch.save(res, os.path.join(out_dir, "forward_step.pt"))
