"""Plugin surface — the drop-in for `import long_vita_megatron.megatron_adaptor`
(M/megatron_adaptor.py:149-174).  Importing this module registers, through the same
`register_patch` / `apply_patches` API, HIP-backed replacements for exactly the targets the
reference patches on the prefill path:

  megatron.core.transformer.dot_product_attention.DotProductAttention.forward   (M/megatron_adaptor.py:21-22, wrapper)
  megatron.core.models.common.embeddings.language_model_embedding.LanguageModelEmbedding   (:93-94)
  megatron.core.tensor_parallel.layers.ColumnParallelLinear                                  (:105-106)
  megatron.core.models.common.embeddings.rotary_pos_embedding.apply_rotary_pos_emb           (K8; the reference
        leaves RotaryEmbedding unpatched, :102-103, and relies on apex's fused kernel)
  megatron.inference.text_generation.generation.generate_tokens_probs_and_return_on_first_stage's helpers
        (get_batch_on_this_cp_rank / sync_output, :141-147) via long_vita_amd.generation

Megatron-LM is not installable in the build container, so the registration is
guarded: without `megatron` the module is importable and `PATCHES` lists what would be applied;
the standalone driver (gpt_vl_model.GPTVLModel, vision.MegatronVisionModel) runs the same kernels.
"""
from __future__ import annotations

import importlib.util
from functools import wraps

from .patch_utils import MindSpeedPatchesManager as aspm


def dot_product_attention_forward_wrapper(fn):
    """Same shape as the reference's wrapper (M/core/transformer/dot_product_attention.py:151-153):
    receives Megatron's DotProductAttention.forward and returns the HIP-backed forward."""
    from .dot_product_attention import DotProductAttention as HipAttention

    @wraps(fn)
    def wrapper(self, query, key, value, attention_mask, attn_mask_type=None, packed_seq_params=None):
        impl = getattr(self, "_vita_hip_impl", None)
        if impl is None:
            np_ = self.num_attention_heads_per_partition
            ng = self.num_query_groups_per_partition
            hn = self.hidden_size_per_attention_head
            # ViT layers are non-causal (dot_product_attention.py:312-329); Megatron marks them with
            # AttnMaskType.no_mask / padding, the LLM with AttnMaskType.causal
            causal = "causal" in str(getattr(self, "attn_mask_type", attn_mask_type)).lower()
            impl = HipAttention(np_, ng, hn, causal=causal)
            self._vita_hip_impl = impl
        return impl.forward(query, key, value, attention_mask, attn_mask_type, packed_seq_params)

    return wrapper


def _targets():
    from .language_model_embedding import LanguageModelEmbedding
    from .layers import ColumnParallelLinear
    from .rotary_pos_embedding import apply_rotary_pos_emb
    from . import generation
    return [
        ("megatron.core.transformer.dot_product_attention.DotProductAttention.forward",
         dot_product_attention_forward_wrapper),
        ("megatron.core.models.common.embeddings.language_model_embedding.LanguageModelEmbedding",
         LanguageModelEmbedding),
        ("megatron.core.tensor_parallel.layers.ColumnParallelLinear", ColumnParallelLinear),
        ("megatron.core.models.common.embeddings.rotary_pos_embedding.apply_rotary_pos_emb", apply_rotary_pos_emb),
        ("megatron.inference.text_generation.generation.get_batch_on_this_cp_rank",
         generation.get_batch_on_this_cp_rank),
        ("megatron.inference.text_generation.generation.sync_output", generation.sync_output),
    ]


PATCHES = [name for name, _ in _targets()]


def exe_adaptation(create_dummy: bool = False) -> bool:
    """Register + apply.  Returns False (and patches nothing) when Megatron-LM is absent."""
    if not create_dummy:
        try:
            if importlib.util.find_spec("megatron") is None:
                return False
        except ValueError:          # a half-initialised stub module without __spec__
            return False
    for name, obj in _targets():
        aspm.register_patch(name, obj, create_dummy=create_dummy)
    aspm.apply_patches()
    return True


APPLIED = exe_adaptation()
