"""Host side of long_vita_amd/hf_adaptor.py (no GPU): the config mapping and the loader's argument checks."""
import json
import os

import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_configs_from_the_reference_config_json_fields():
    """config_14B.json's fields (H/models/long_vita_qwen2_intern/config_14B.json:1-56; the small copy under oracle/ differs in sizes only)
    -> GPTConfig / VisionConfig; a dict and an attribute object give the same result."""
    import types
    from long_vita_amd import gpt_vl_model, hf_adaptor, vision
    full = dict(hidden_size=5120, num_attention_heads=40, num_key_value_heads=8, num_hidden_layers=48, intermediate_size=13824,
                vocab_size=152064, rms_norm_eps=1e-6, rope_theta=1000000.0,
                visual=dict(hidden_size=1024, num_attention_heads=16, num_hidden_layers=24, intermediate_size=4096, patch_size=14,
                            image_size=448, layer_norm_eps=1e-6, qk_normalization=False, norm_type="layer_norm", hidden_act="gelu"))
    g, v = hf_adaptor.configs_from_hf(full)
    assert g == gpt_vl_model.GPTConfig() and v == vision.VisionConfig()
    obj = types.SimpleNamespace(**{**full, "visual": types.SimpleNamespace(**full["visual"])})
    assert hf_adaptor.configs_from_hf(obj) == (g, v)
    small = json.load(open(os.path.join(ROOT, "oracle", "hf_long_vita_small_config.json")))
    gs, vs = hf_adaptor.configs_from_hf(small)
    assert (gs.hidden, gs.heads, gs.kv_groups, gs.head_dim, gs.ffn, gs.vocab, gs.num_layers) == (1024, 8, 2, 128, 2816, 1024, 2)
    assert vs.num_layers == 2 and vs.llm_hidden == 1024
    with pytest.raises(NotImplementedError):
        hf_adaptor.configs_from_hf({**full, "visual": {**full["visual"], "qk_normalization": True}})
    text_only = {k: v_ for k, v_ in full.items() if k != "visual"}
    assert hf_adaptor.configs_from_hf(text_only)[1] is None


def test_from_pretrained_refuses_other_dtypes_and_has_no_fallback(tmp_path):
    from long_vita_amd import hf_adaptor
    with pytest.raises(ValueError, match="bf16"):
        hf_adaptor.LongVITAForCausalLM.from_pretrained(str(tmp_path), torch_dtype=torch.float32)
    with pytest.raises(FileNotFoundError):
        hf_adaptor.LongVITAForCausalLM.from_pretrained(str(tmp_path), torch_dtype=torch.bfloat16)
