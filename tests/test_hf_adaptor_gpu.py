"""The transformers entry point on the HIP path (long_vita_amd/hf_adaptor.py).

  * BASELINE config 1's shape — a [1, 2048] text-only prefill, here through 4 FULL-WIDTH decoder layers (hidden 5120, 40:8 heads, FFN 13824,
    vocabulary 152064) — against `transformers.Qwen2ForCausalLM` in fp32 on the host (the class the reference's LongVITAForCausalLM derives
    from, H/models/long_vita_qwen2_intern/modeling_long_vita.py:238-246): logits of sampled rows, the cached decode steps, greedy `generate`.
  * One image + text through `forward(input_ids, images, image_indices)` and `generate(inputs=, images=, image_indices=)` against
    tests/golden/hf_long_vita.pt — the reference's own InternVisionModel + ResamplerProjector, transformers' Qwen2ForCausalLM and the scatter of
    modeling_long_vita.py:137-147, fp32 on the CPU (oracle/make_golden.py:golden_hf_long_vita); the scatter itself bit for bit.
Tolerances: bf16 path vs fp32 reference, the "chain" scale of DESIGN.md section 2 (2 - 4 layers: ~1e-2); each measured value is recorded.
"""
import os
import sys

import pytest
import torch

pytestmark = pytest.mark.gpu

from conftest import load_golden, tol  # noqa: E402

DEV = "cuda"


def rel_l2(a, b):
    a, b = a.float().cpu(), b.float().cpu()
    return float((a - b).norm() / b.norm())


def test_text_only_2k_prefill_four_full_width_layers_vs_transformers_qwen2():
    import transformers
    from long_vita_amd import hf_adaptor
    from oracle import llm as ollm
    lc = ollm.LLMConfig(num_layers=4)
    p = ollm.init_llm_params(lc, seed=21)
    sd = ollm.to_hf_state_dict(p, lc)
    config = dict(hidden_size=lc.hidden, num_attention_heads=lc.heads, num_key_value_heads=lc.kv_groups, num_hidden_layers=lc.num_layers,
                  intermediate_size=lc.ffn, vocab_size=lc.vocab, rms_norm_eps=lc.eps, rope_theta=lc.rope_theta, eos_token_id=151645,
                  tie_word_embeddings=False, use_cache=True)
    model = hf_adaptor.LongVITAForCausalLM.from_state_dict(config, sd, device=DEV).eval()
    S, new = 2048, 4
    g = torch.Generator().manual_seed(22)
    ids = torch.randint(0, 151643, (1, S), generator=g)
    out = model(input_ids=ids.to(DEV))                                                     # every row, as num_logits_to_keep = 0 asks
    assert out.logits.shape == (1, S, lc.vocab) and out.past_key_values is not None
    last = model(input_ids=ids.to(DEV), num_logits_to_keep=1, use_cache=False)
    assert last.logits.shape == (1, 1, lc.vocab) and last.past_key_values is None
    gen = model.generate(inputs=ids, max_new_tokens=new)
    assert gen.shape == (1, S + new) and torch.equal(gen[:, :S].cpu(), ids)
    # cached steps through forward(past_key_values=...), teacher-forced on what generate() produced
    step_logits, cache = [], out.past_key_values
    for j in range(new - 1):
        o = model(input_ids=gen[:, S + j: S + j + 1], past_key_values=cache)
        step_logits.append(o.logits[0, -1].float().cpu())
        cache = o.past_key_values
    assert cache.get_seq_length() == S + new - 1

    qcfg = transformers.Qwen2Config(vocab_size=lc.vocab, hidden_size=lc.hidden, intermediate_size=lc.ffn, num_hidden_layers=lc.num_layers,
                                    num_attention_heads=lc.heads, num_key_value_heads=lc.kv_groups, rms_norm_eps=lc.eps, rope_theta=lc.rope_theta,
                                    max_position_embeddings=4096, tie_word_embeddings=False, attention_dropout=0.0)
    ref_model = transformers.Qwen2ForCausalLM(qcfg).eval().float()
    missing = ref_model.load_state_dict({k: v.float() for k, v in sd.items()}, strict=False)
    assert not missing.unexpected_keys, missing
    with torch.no_grad():
        ref = ref_model(input_ids=gen[:, :-1].cpu()).logits[0]                             # [S + new - 1, V] fp32: prompt rows + the teacher-forced steps
    rows = sorted({0, 1, 63, 64, 255, 256, 1023, 1024, S - 2, S - 1} | {int(x) for x in torch.randint(0, S, (54,), generator=g)})
    tol("prefill logits, 64 rows", rel_l2(out.logits[0, rows], ref[rows]), 3.8e-2)          # measured 2.54e-2 (4 full-width bf16 layers vs fp32)
    tol("last row, num_logits_to_keep=1", rel_l2(last.logits[0, 0], ref[S - 1]), 3.8e-2)
    assert rel_l2(last.logits[0, 0], out.logits[0, S - 1]) < 5e-3
    for j, sl in enumerate(step_logits):
        tol("cached decode step logits", rel_l2(sl, ref[S + j]), 3.8e-2)
    # greedy tokens: equal wherever the reference's own top-2 margin is clear of bf16 rounding
    for j in range(new):
        top = torch.topk(ref[S - 1 + j], 2)
        if float(top.values[0] - top.values[1]) > 0.15:
            assert int(gen[0, S + j]) == int(top.indices[0]), (j, gen[0, S:], top)
        else:
            break


def test_image_request_forward_and_generate_vs_the_reference_fixture():
    from long_vita_amd import hf_adaptor
    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    from oracle import make_golden as mg
    g = load_golden("hf_long_vita.pt")
    config, sd, ids, images, idx, lc, vc = mg.hf_long_vita_case(g["case"])
    model = hf_adaptor.LongVITAForCausalLM.from_state_dict(config, sd, device=DEV).eval()
    assert model.model.cfg.hidden == lc.hidden and model.model.external_feature_model.cfg.num_layers == vc.num_layers
    out = model(input_ids=ids, images=images, image_indices=idx, use_cache=False)
    assert out.logits.shape == (1, ids.shape[1], lc.vocab)
    tol("image request: logits rows ::7", rel_l2(out.logits[0, ::7], g["logits_rows"]), 2.0e-2)
    tol("image request: last row", rel_l2(out.logits[0, -1], g["logits_last"]), 2.0e-2)
    # the parts: projected features, and the scatter of modeling_long_vita.py:137-147 bit for bit on the same bf16 tensors
    feats = model.model.external_feature_model(images=images.to(DEV))                       # [1, 256, h]
    tol("image request: projected features", rel_l2(feats[0, ::8, ::8], g["image_embeds_sub"]), 1.5e-2)
    emb = model.model.embedding(input_ids=ids.to(DEV), position_ids=None, external_feature_dict={"features": feats, "indices": idx.to(DEV)})
    want = model.model.p["embed"][ids.to(DEV)].clone()                                      # inputs_embeds = self.embed_tokens(input_ids); .clone()
    indices_b, indices_s = idx.to(DEV).unbind(dim=0)
    want[indices_b.view(-1), indices_s.view(-1)] = feats.view(-1, feats.shape[-1])
    assert torch.equal(emb.transpose(0, 1), want)
    tol("image request: scattered embeddings", rel_l2(emb.transpose(0, 1)[0, ::5, ::16], g["embeds_sub"]), 1.5e-2)
    # greedy generate: the reference's tokens while its own top-2 margin is clear of bf16 rounding
    n = g["case"]["new_tokens"]
    gen = model.generate(inputs=ids, images=images, image_indices=idx, max_new_tokens=n)
    assert gen.shape == (1, ids.shape[1] + n)
    compared = 0
    for j in range(n):
        if float(g["top2_gaps"][j]) < 0.1:
            break
        assert int(gen[0, ids.shape[1] + j]) == int(g["generated"][j]), (j, gen[0, ids.shape[1]:], g["generated"])
        compared += 1
    assert compared >= 2
    # an end-of-sequence id stops the loop behind it
    eos = int(gen[0, ids.shape[1] + 1])
    short = model.generate(inputs=ids, images=images, image_indices=idx, max_new_tokens=n, eos_token_id=eos)
    assert short.shape[1] <= ids.shape[1] + 2 and int(short[0, -1]) == eos


def test_argument_errors_match_the_class_surface():
    from long_vita_amd import hf_adaptor
    from oracle import llm as ollm
    lc = ollm.LLMConfig(num_layers=1, hidden=1024, heads=8, kv_groups=2, ffn=2816, vocab=1024)
    sd = ollm.to_hf_state_dict(ollm.init_llm_params(lc, seed=3), lc)
    config = dict(hidden_size=1024, num_attention_heads=8, num_key_value_heads=2, num_hidden_layers=1, intermediate_size=2816, vocab_size=1024)
    model = hf_adaptor.LongVITAForCausalLM.from_state_dict(config, sd, device=DEV)
    ids = torch.randint(0, 1024, (1, 128))
    with pytest.raises(ValueError, match="exactly one of input_ids or inputs_embeds"):
        model()
    with pytest.raises(ValueError, match="batch 1"):
        model(input_ids=torch.cat([ids, ids]))
    with pytest.raises(ValueError, match="padded attention masks"):
        model(input_ids=ids, attention_mask=torch.tensor([[0] + [1] * 127]))
    with pytest.raises(NotImplementedError):
        model(input_ids=ids, labels=ids)
    with pytest.raises(ValueError, match="without vision weights"):
        model(input_ids=ids, images=torch.zeros(1, 3, 448, 448), image_indices=torch.zeros(2, 1, 256, dtype=torch.long))
    # inputs_embeds (what the reference's forward builds at :137) gives the logits of the ids it came from
    a = model(input_ids=ids, use_cache=False).logits
    emb = model.model.p["embed"][ids.to(DEV)]
    b = model(inputs_embeds=emb, use_cache=False).logits
    assert torch.equal(a, b)


def test_from_pretrained_reads_a_checkpoint_directory(tmp_path):
    """`AutoModelForCausalLM.from_pretrained(model_path, trust_remote_code=True, device_map="auto", torch_dtype=torch.bfloat16,
    attn_implementation="flash_attention_2")` of R/tools/inference_long_vita.py:811-817 on a `*_HF`-layout directory (config.json,
    generation_config.json, sharded safetensors + index): the same logits, bit for bit, as `from_state_dict` on the tensors it was written from."""
    import json
    from safetensors.torch import save_file
    from long_vita_amd import hf_adaptor
    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    from oracle import make_golden as mg
    config, sd, ids, images, idx, lc, vc = mg.hf_long_vita_case()
    names = sorted(sd)
    half = len(names) // 2
    shards = {"model-00001-of-00002.safetensors": names[:half], "model-00002-of-00002.safetensors": names[half:]}
    for fn, keys in shards.items():
        save_file({k: sd[k].contiguous() for k in keys}, str(tmp_path / fn))
    json.dump({"metadata": {}, "weight_map": {k: fn for fn, keys in shards.items() for k in keys}}, open(tmp_path / "model.safetensors.index.json", "w"))
    json.dump(config, open(tmp_path / "config.json", "w"))
    json.dump({"max_new_tokens": 3, "do_sample": False, "eos_token_id": [2, 5]}, open(tmp_path / "generation_config.json", "w"))
    a = hf_adaptor.LongVITAForCausalLM.from_pretrained(str(tmp_path), trust_remote_code=True, device_map="auto", torch_dtype=torch.bfloat16,
                                                       attn_implementation="flash_attention_2").eval()
    b = hf_adaptor.LongVITAForCausalLM.from_state_dict(config, sd, device=DEV)
    la = a(input_ids=ids, images=images, image_indices=idx, num_logits_to_keep=4, use_cache=False).logits
    lb = b(input_ids=ids, images=images, image_indices=idx, num_logits_to_keep=4, use_cache=False).logits
    assert la.shape == (1, 4, lc.vocab) and torch.equal(la, lb)
    assert a.generation_config.max_new_tokens == 3 and a.generation_config.eos_token_id == [2, 5]
    gen = a.generate(inputs=ids, images=images, image_indices=idx)
    assert ids.shape[1] < gen.shape[1] <= ids.shape[1] + 3
