"""Checkpoint converters: transformers <-> Megatron layouts, tensor-parallel shard merging,
and the two on-disk formats, all on CPU (pure tensor slicing, compared for exact equality)."""
import json
import os

import pytest
import torch

from long_vita_amd import checkpoint as ck
from oracle import llm as ollm, vit as ovit


def _tree_equal(a, b):
    if isinstance(a, dict):
        return a.keys() == b.keys() and all(_tree_equal(a[k], b[k]) for k in a)
    if isinstance(a, list):
        return len(a) == len(b) and all(_tree_equal(x, y) for x, y in zip(a, b))
    return torch.equal(a, b)


CFG = ollm.LLMConfig(num_layers=2, hidden=256, heads=8, kv_groups=2, head_dim=32, ffn=512, vocab=320)


def test_hf_qwen2_weights_convert_and_reproduce_transformers_logits():
    from transformers import Qwen2Config, Qwen2ForCausalLM
    torch.manual_seed(5)
    hf = Qwen2ForCausalLM(Qwen2Config(vocab_size=320, hidden_size=256, intermediate_size=512, num_hidden_layers=2,
                                      num_attention_heads=8, num_key_value_heads=2, rms_norm_eps=1e-6, rope_theta=1e6,
                                      max_position_embeddings=4096, tie_word_embeddings=False,
                                      attn_implementation="eager")).eval()
    with torch.no_grad():
        for prm in hf.parameters():                      # biases / norms are zeros / ones by default: randomise
            prm.copy_(torch.randn_like(prm) * 0.05)
    p = ck.hf_llm_to_params(hf.state_dict(), CFG)
    tokens = torch.randint(0, 320, (1, 80), generator=torch.Generator().manual_seed(1))
    with torch.no_grad():
        ref = hf(tokens).logits
    out = ollm.prefill_logits(tokens, p, CFG, range(80))
    torch.testing.assert_close(out, ref, rtol=2e-4, atol=2e-4)
    # exact inverse of the restated reference converter
    assert _tree_equal(ck.hf_llm_to_params(ollm.to_hf_state_dict(p, CFG), CFG), p)


def _mcore_names(p):
    sd = {"embedding.word_embeddings.weight": p["embed"], "decoder.final_layernorm.weight": p["final_ln"],
          "output_layer.weight": p["lm_head"]}
    for i, lp in enumerate(p["layers"]):
        pre = f"decoder.layers.{i}."
        sd[pre + "self_attention.linear_qkv.layer_norm_weight"] = lp["ln1"]
        sd[pre + "self_attention.linear_qkv.weight"] = lp["qkv_w"]
        sd[pre + "self_attention.linear_qkv.bias"] = lp["qkv_b"]
        sd[pre + "self_attention.linear_proj.weight"] = lp["o_w"]
        sd[pre + "mlp.linear_fc1.layer_norm_weight"] = lp["ln2"]
        sd[pre + "mlp.linear_fc1.weight"] = lp["fc1_w"]
        sd[pre + "mlp.linear_fc2.weight"] = lp["fc2_w"]
        sd[pre + "self_attention.linear_qkv._extra_state"] = None          # TE writes these
    return sd


def _megatron_tp_split(sd, tp):
    """What Megatron's Column/RowParallelLinear + VocabParallelEmbedding hold on each TP rank."""
    shards = [dict() for _ in range(tp)]
    for name, t in sd.items():
        for r in range(tp):
            if t is None:
                shards[r][name] = None
            elif name.endswith("mlp.linear_fc1.weight"):                      # SwiGLU: gate and up are split separately
                gte, up = t.chunk(2, dim=0)
                shards[r][name] = torch.cat([gte.chunk(tp, 0)[r], up.chunk(tp, 0)[r]])
            elif name.endswith(("linear_qkv.weight", "linear_qkv.bias", "word_embeddings.weight", "output_layer.weight")):
                shards[r][name] = t.chunk(tp, dim=0)[r].clone()
            elif name.endswith(("linear_proj.weight", "linear_fc2.weight")):
                shards[r][name] = t.chunk(tp, dim=1)[r].clone()
            else:
                shards[r][name] = t.clone()
    return shards


@pytest.mark.parametrize("tp", [1, 2])
def test_mcore_checkpoint_directory_roundtrip(tmp_path, tp):
    p = ollm.init_llm_params(CFG, seed=3, dtype=torch.bfloat16)
    shards = _megatron_tp_split(_mcore_names(p), tp)
    root = tmp_path / "ckpt"
    for r, s in enumerate(shards):
        d = root / "iter_0000042" / f"mp_rank_{r:02d}"
        os.makedirs(d)
        torch.save({"model": s, "iteration": 42}, d / "model_optim_rng.pt")
    (root / "latest_checkpointed_iteration.txt").write_text("42")
    merged = ck.load_mcore_checkpoint(str(root))
    assert _tree_equal(ck.mcore_llm_to_params(merged, CFG), p)


@pytest.mark.parametrize("form", ["combined", "split"])
def test_tp2_merge_with_the_references_vision_prefixes(form):
    """at TP = 2 the GELU fc1 of the ViT (`external_feature_model.vit.*`, M/pretrain_long_vita.py:381) and of
    the projector (`external_feature_model.vision_projection.*`, :436) is plain column-parallel — only the decoder's fc1 is a
    [gate; up] pair.  Combined checkpoint and the two halves M/ckpt_split_llm_and_vit.py writes (`vit.*`, `vision_projection.*`)."""
    g = torch.Generator().manual_seed(5)
    rn = lambda *s: torch.randn(*s, generator=g)                                       # noqa: E731
    full = {"decoder.layers.0.mlp.linear_fc1.weight": rn(16, 4), "decoder.layers.0.mlp.linear_fc2.weight": rn(4, 8),
            "external_feature_model.vit.conv1.weight": rn(6, 3, 2, 2),
            "external_feature_model.vit.decoder.layers.0.mlp.linear_fc1.weight": rn(12, 6),
            "external_feature_model.vit.decoder.layers.0.mlp.linear_fc1.bias": rn(12),
            "external_feature_model.vit.decoder.layers.0.mlp.linear_fc2.weight": rn(6, 12),
            "external_feature_model.vision_projection.encoder.linear_fc1.weight": rn(8, 24),
            "external_feature_model.vision_projection.encoder.linear_fc2.weight": rn(4, 8)}

    def split(sd):
        shards = [dict(), dict()]
        for name, t in sd.items():
            for r in range(2):
                if name.endswith("linear_fc1.weight") and name.startswith("decoder."):   # SwiGLU: gate / up split apart
                    a, b = t.chunk(2, 0)
                    shards[r][name] = torch.cat([a.chunk(2, 0)[r], b.chunk(2, 0)[r]])
                elif name.endswith(("linear_fc1.weight", "linear_fc1.bias")):
                    shards[r][name] = t.chunk(2, 0)[r].clone()
                elif name.endswith("linear_fc2.weight"):
                    shards[r][name] = t.chunk(2, 1)[r].clone()
                else:
                    shards[r][name] = t.clone()
        return shards

    if form == "combined":
        merged = ck.merge_tp_shards(split(full))
        assert set(merged) == set(full) and all(torch.equal(merged[k], full[k]) for k in full)
    else:
        llm, vit = ck.split_llm_and_vit(full)
        assert all(k.startswith(("vit.", "vision_projection.")) for k in vit)
        for part in (llm, vit):
            shards = split(part)
            merged = ck.merge_tp_shards(shards)
            assert all(torch.equal(merged[k], part[k]) for k in part), form


def test_hf_safetensors_sharded_directory(tmp_path):
    from safetensors.torch import save_file
    p = ollm.init_llm_params(CFG, seed=4, dtype=torch.bfloat16)
    sd = {k: v.contiguous() for k, v in ollm.to_hf_state_dict(p, CFG).items()}
    keys = sorted(sd)
    half = len(keys) // 2
    files = {"model-00001-of-00002.safetensors": keys[:half], "model-00002-of-00002.safetensors": keys[half:]}
    for fn, ks in files.items():
        save_file({k: sd[k] for k in ks}, str(tmp_path / fn))
    (tmp_path / "model.safetensors.index.json").write_text(json.dumps(
        {"weight_map": {k: fn for fn, ks in files.items() for k in ks}}))
    assert _tree_equal(ck.hf_llm_to_params(ck.load_hf_safetensors(str(tmp_path)), CFG), p)


def test_intern_vit_layouts():
    vcfg = ovit.ViTConfig(num_layers=2)
    p = ovit.init_vit_params(vcfg, seed=9, dtype=torch.bfloat16)
    # HF names (modeling_intern_vit.py) from the Megatron-layout params, with the oracle's inverse permutation
    sd = {"embeddings.class_embedding": p["cls"], "embeddings.patch_embedding.weight": p["conv_w"],
          "embeddings.patch_embedding.bias": p["conv_b"], "embeddings.position_embedding": p["pos"][None],
          "proj.pre_proj_layernorm.weight": p["proj_ln_w"], "proj.pre_proj_layernorm.bias": p["proj_ln_b"],
          "proj.mlp.0.weight": p["proj_fc1"], "proj.mlp.2.weight": p["proj_fc2"]}
    for i, lp in enumerate(p["layers"]):
        pre = f"encoder.layers.{i}."
        sd[pre + "attn.qkv.weight"] = ovit.megatron_qkv_to_hf(lp["qkv_w"], 16, 64)
        sd[pre + "attn.qkv.bias"] = ovit.megatron_qkv_to_hf(lp["qkv_b"], 16, 64)
        for a, b in [("attn.proj.weight", "proj_w"), ("attn.proj.bias", "proj_b"), ("mlp.fc1.weight", "fc1_w"),
                     ("mlp.fc1.bias", "fc1_b"), ("mlp.fc2.weight", "fc2_w"), ("mlp.fc2.bias", "fc2_b"),
                     ("norm1.weight", "ln1_w"), ("norm1.bias", "ln1_b"), ("norm2.weight", "ln2_w"),
                     ("norm2.bias", "ln2_b"), ("ls1", "ls1"), ("ls2", "ls2")]:
            sd[pre + a] = lp[b]
    assert _tree_equal(ck.hf_vit_to_params(sd, vcfg, projector_prefix="proj."), p)
    # the permutation equals the reference converter's `indices` gather (ckpt_converter_intern_vit.py:54-66)
    kv, hid, heads = 64, 1024, 16
    idx = torch.cat([torch.arange(j * hid + i * kv, j * hid + (i + 1) * kv) for i in range(heads) for j in range(3)])
    w = sd["encoder.layers.0.attn.qkv.weight"]
    assert torch.equal(ck.hf_qkv_to_megatron(w, heads, kv), w[idx])
    # Megatron names + TP = 2 chunking as the reference converter writes them (:150-158), then merge
    mg = {"class_token": p["cls"].expand(1, 1, -1), "position_embeddings.weight": p["pos"], "conv1.weight": p["conv_w"],
          "conv1.bias": p["conv_b"]}
    for i, lp in enumerate(p["layers"]):
        pre = f"decoder.layers.{i}."
        for a, b in [("self_attention.linear_qkv.weight", "qkv_w"), ("self_attention.linear_qkv.bias", "qkv_b"),
                     ("self_attention.linear_proj.weight", "proj_w"), ("self_attention.linear_proj.bias", "proj_b"),
                     ("self_attention.linear_qkv.layer_norm_weight", "ln1_w"),
                     ("self_attention.linear_qkv.layer_norm_bias", "ln1_b"), ("mlp.linear_fc1.weight", "fc1_w"),
                     ("mlp.linear_fc1.bias", "fc1_b"), ("mlp.linear_fc2.weight", "fc2_w"), ("mlp.linear_fc2.bias", "fc2_b"),
                     ("mlp.linear_fc1.layer_norm_weight", "ln2_w"), ("mlp.linear_fc1.layer_norm_bias", "ln2_b"),
                     ("ls1", "ls1"), ("ls2", "ls2")]:
            mg[pre + a] = lp[b]
    chunk0 = ("linear_qkv.weight", "linear_qkv.bias", "linear_fc1.weight", "linear_fc1.bias")
    chunk1 = ("linear_proj.weight", "linear_fc2.weight")
    shards = [{k: (v.chunk(2, 0)[r] if k.endswith(chunk0) else v.chunk(2, 1)[r] if k.endswith(chunk1) else v)
               for k, v in mg.items()} for r in range(2)]
    merged = ck.merge_tp_shards(shards, swiglu_fc1=False)
    got = ck.mcore_vit_to_params(merged, vcfg)
    want = {k: v for k, v in p.items() if not k.startswith("proj_")}
    assert _tree_equal(got, want)


def test_vit_loader_reads_what_the_references_converter_writes(tmp_path):
    """L/ckpt_converter_intern_vit.py:convert was run on a marker state dict (tensor-parallel size 2, --use-te); the fixture
    (converters.pt) keeps, per written tensor and rank, which source rows / columns it holds.  Rebuild those two shard files
    from a random transformers state dict, read them with load_mcore_checkpoint + mcore_vit_to_params, and get exactly what
    hf_vit_to_params makes of the transformers names — names, the per-head [q|k|v] gather, chunk dims and the merge all agree
    with the reference's own converter."""
    from conftest import load_golden
    from oracle.make_golden import VIT_CONVERT_SHAPES
    g = load_golden("converters.pt")["vit"]
    gen = torch.Generator().manual_seed(21)
    hf = {k: torch.randn(*shape, generator=gen) for k, shape in VIT_CONVERT_SHAPES.items()}

    def two_d(t):
        while t.dim() > 1 and t.shape[0] == 1:
            t = t[0]
        return t.reshape(t.shape[0], -1)

    root = tmp_path / "ckpt"
    assert len(g["ranks"]) == 2
    for r, entry in enumerate(g["ranks"]):
        shard = {}
        for name, e in entry.items():
            if e is None:
                shard[name] = None                                    # TE _extra_state placeholder, as written
                continue
            src = two_d(hf[e["src"]])
            shard[name] = src[torch.tensor(e["rowmap"])][:, torch.tensor(e["colmap"])].reshape(e["shape"]).clone()
        d = root / "iter_0000001" / f"mp_rank_{r:02d}"
        os.makedirs(d)
        torch.save({"model": shard}, d / "model_optim_rng.pt")
    (root / "latest_checkpointed_iteration.txt").write_text("1")
    vcfg = ovit.ViTConfig(num_layers=1)
    got = ck.mcore_vit_to_params(ck.load_mcore_checkpoint(str(root)), vcfg)
    want = ck.hf_vit_to_params(hf, vcfg)
    assert _tree_equal(got, want)


def test_llm_converter_matches_the_references_hf2mcore():
    """R/tools/hf2mcore_long_vita.py:convert_checkpoint_from_transformers_to_megatron (source executed on stand-in module
    trees, fixture converters.pt): the Megatron language-model state dict it fills from a transformers state dict is, tensor for
    tensor, what hf_llm_to_params builds from the same state dict and what mcore_llm_to_params reads back."""
    from conftest import load_golden
    from oracle.make_golden import LLM_CONVERT_DIMS, llm_convert_hf_state
    g = load_golden("converters.pt")["llm"]
    assert g["dims"] == LLM_CONVERT_DIMS
    dm = g["dims"]
    cfg = ollm.LLMConfig(num_layers=dm["layers"], hidden=dm["hidden"], heads=dm["heads"], kv_groups=dm["groups"],
                         head_dim=dm["head_dim"], ffn=dm["ffn"], vocab=dm["vocab"])
    want = ck.mcore_llm_to_params(g["megatron_state"], cfg)
    got = ck.hf_llm_to_params(llm_convert_hf_state(), cfg)
    assert _tree_equal(got, want)
    assert not torch.equal(got["lm_head"], got["embed"])             # --untie-embeddings-and-output-weights


def test_checkpoint_scripts_of_the_reference(tmp_path):
    """M/ckpt_convert_modellink_to_megatron_with_te.py and M/ckpt_split_llm_and_vit.py were run on a two-rank directory whose
    tensors are their own key index (fixture ckpt_scripts.pt).  split_llm_and_vit makes the same two dictionaries; the
    local-spec -> TE renames are exactly the alternative names mcore_llm_to_params accepts; a directory in Megatron's PP > 1
    naming with only stage 000 loads, other stages are refused."""
    from conftest import load_golden
    g = load_golden("ckpt_scripts.pt")
    assert g["tracker"] == "7"
    for r in range(2):
        sd = {k: torch.tensor([100 * r + i]) for i, k in enumerate(g["keys"])}
        llm, vit = ck.split_llm_and_vit(sd)
        assert {k: int(v) for k, v in llm.items()} == g["llm"][f"iter_0000007/mp_rank_{r:02d}_000/model_optim_rng.pt"]
        assert {k: int(v) for k, v in vit.items()} == g["vit"][f"iter_0000007/mp_rank_{r:02d}/model_optim_rng.pt"]
        ren = g["renamed"][f"iter_0000007/mp_rank_{r:02d}_000/model_optim_rng.pt"]
        moved = {k: next(k2 for k2, v2 in ren.items() if v2 == int(v)) for k, v in sd.items() if k not in ren}
        assert moved == {"decoder.layers.0.input_layernorm.weight": "decoder.layers.0.self_attention.linear_qkv.layer_norm_weight",
                         "decoder.layers.0.pre_mlp_layernorm.weight": "decoder.layers.0.mlp.linear_fc1.layer_norm_weight"}
    # both spellings read the same parameters
    cfg1 = ollm.LLMConfig(num_layers=1, hidden=CFG.hidden, heads=CFG.heads, kv_groups=CFG.kv_groups, head_dim=CFG.head_dim,
                          ffn=CFG.ffn, vocab=CFG.vocab)
    p = ollm.init_llm_params(cfg1, seed=6, dtype=torch.bfloat16)
    te = _mcore_names(p)
    local = {moved_back: v for moved_back, v in ((next((a for a, b in moved.items() if b == k), k), v) for k, v in te.items())}
    assert set(local) != set(te) and _tree_equal(ck.mcore_llm_to_params(local, cfg1), ck.mcore_llm_to_params(te, cfg1))
    # Megatron's directory names when the pipeline rank is part of them
    root = tmp_path / "ckpt"
    d = root / "iter_0000007" / "mp_rank_00_000"
    os.makedirs(d)
    torch.save({"model": te}, d / "model_optim_rng.pt")
    (root / "latest_checkpointed_iteration.txt").write_text("7")
    assert _tree_equal(ck.mcore_llm_to_params(ck.load_mcore_checkpoint(str(root)), cfg1), p)
    os.makedirs(root / "iter_0000007" / "mp_rank_00_001")
    with pytest.raises(NotImplementedError):
        ck.load_mcore_checkpoint(str(root))
