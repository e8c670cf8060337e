"""The drop-in boundary under a (stand-in) Megatron, on the GPU (VERDICT r1 "next round" item 2):
one decoder layer built by `build_module(get_gpt_layer_*_spec(), config=, layer_number=)` — Megatron's construction path, the
spec builders taken from the dotted names the adaptor patched (M/megatron_adaptor.py:81-88) — whose leaves are this package's
HIP-backed nn.Modules; forward AND backward (torch autograd through autograd_fns) against the oracle layer
(oracle.llm.decoder_layer: the TE layer spec order of M/core/models/gpt/gpt_layer_specs.py:35-49) and torch autograd over it.
RoPE arrives the way Megatron passes it: fp32 `freqs` [s, 1, 1, d] (rotary_pos_embedding.py:232-259)."""
import sys

import pytest
import torch

pytestmark = pytest.mark.gpu

import dummy_megatron as dm  # noqa: E402
from oracle import attention as oattn, glue, llm as ollm  # noqa: E402

from conftest import tol  # noqa: E402

DEV = "cuda"
CFG = dict(num_layers=1, hidden=1024, heads=8, kv_groups=2, head_dim=128, ffn=2816, vocab=512)


def rel_l2(a, b):
    a, b = a.float().cpu(), b.float().cpu()
    return float((a - b).norm() / (b.norm() + 1e-30))


def _record(name, errs):
    from conftest import record_parity
    record_parity("boundary/" + name, **{k: float(v) for k, v in errs.items()})


@pytest.fixture()
def megatron():
    import long_vita_amd.megatron_adaptor as ad
    from long_vita_amd.patch_utils import MindSpeedPatchesManager as aspm
    from long_vita_amd import ops
    ops._L.load(allow_build=False)
    aspm.patches_info = {}
    names = dm.install()
    assert ad.exe_adaptation(create_dummy=True)
    yield sys.modules["megatron.core.models.gpt.gpt_layer_specs"]
    dm.uninstall(names)
    aspm.patches_info = {}


def _load(layer, lp, te: bool):
    sd = {"self_attention.linear_qkv.weight": lp["qkv_w"], "self_attention.linear_qkv.bias": lp["qkv_b"],
          "self_attention.linear_proj.weight": lp["o_w"], "mlp.linear_fc1.weight": lp["fc1_w"], "mlp.linear_fc2.weight": lp["fc2_w"]}
    if te:
        sd["self_attention.linear_qkv.layer_norm_weight"], sd["mlp.linear_fc1.layer_norm_weight"] = lp["ln1"], lp["ln2"]
    else:
        sd["input_layernorm.weight"], sd["pre_mlp_layernorm.weight"] = lp["ln1"], lp["ln2"]
    layer.load_state_dict({k: v.to(DEV) for k, v in sd.items()})                      # Megatron checkpoint names, no _extra_state
    return sd


@pytest.mark.parametrize("spec", ["te", "local"])
def test_decoder_layer_built_by_megatron_matches_the_oracle_forward_and_backward(megatron, spec):
    S = 512
    ocfg = ollm.LLMConfig(**CFG)
    p = ollm.init_llm_params(ocfg, seed=31)
    lp = {k: (v * 1.0 if "ln" not in k else (1.0 + 0.1 * torch.randn(v.shape, generator=torch.Generator().manual_seed(5))).to(v.dtype))
          for k, v in p["layers"][0].items()}                                          # non-trivial norm weights
    mcfg = dm.TransformerConfig(hidden_size=CFG["hidden"], num_attention_heads=CFG["heads"], num_query_groups=CFG["kv_groups"],
                                kv_channels=CFG["head_dim"], ffn_hidden_size=CFG["ffn"])
    builder = megatron.get_gpt_layer_with_transformer_engine_spec if spec == "te" else megatron.get_gpt_layer_local_spec
    layer = dm.build_module(builder(), config=mcfg, layer_number=1)
    assert all(q.is_cuda for q in layer.parameters())                                  # allocated on the current HIP device
    _load(layer, lp, spec == "te")

    g = torch.Generator().manual_seed(32)
    x = (torch.randn(S, 1, CFG["hidden"], generator=g) * 0.5).bfloat16()
    w_out = torch.randn(S, 1, CFG["hidden"], generator=g).bfloat16()
    freqs = glue.rope_emb(S, glue.rope_inv_freq(ocfg.head_dim, ocfg.rope_theta))      # fp32 [s, 1, 1, d], as RotaryEmbedding.forward returns

    # ---- oracle: bf16 rounding chain + torch autograd -----------------------------------------------------------------------
    xo = x.clone().requires_grad_(True)
    lpo = {k: v.clone().requires_grad_(True) for k, v in lp.items()}
    ref, _ = ollm.decoder_layer(xo, lpo, ocfg, freqs, lambda q, k, v: oattn.core_attention(q, k, v, causal=True))
    (ref.float() * w_out.float()).sum().backward()

    # ---- the Megatron-built layer --------------------------------------------------------------------------------------------
    xh = x.to(DEV).requires_grad_(True)
    out, _ = layer(xh, attention_mask=None, rotary_pos_emb=freqs.to(DEV))
    assert out.shape == (S, 1, CFG["hidden"]) and out.dtype == torch.bfloat16
    e_fwd = rel_l2(out, ref)
    tol("forward", e_fwd, 3.7e-03)
    (out.float() * w_out.to(DEV).float()).sum().backward()
    names = {"qkv_w": "self_attention.linear_qkv.weight", "qkv_b": "self_attention.linear_qkv.bias",
             "o_w": "self_attention.linear_proj.weight", "fc1_w": "mlp.linear_fc1.weight", "fc2_w": "mlp.linear_fc2.weight",
             "ln1": "self_attention.linear_qkv.layer_norm_weight" if spec == "te" else "input_layernorm.weight",
             "ln2": "mlp.linear_fc1.layer_norm_weight" if spec == "te" else "pre_mlp_layernorm.weight"}
    params = dict(layer.named_parameters())
    errs = {"dx": rel_l2(xh.grad, xo.grad)}
    for k, n in names.items():
        assert params[n].grad is not None, n
        errs[k] = rel_l2(params[n].grad, lpo[k].grad)
    _record("layer_" + spec, errs)
    tol("worst gradient", max(errs.values()), 8.6e-03)

    # inference call (no autograd): the in-place fast path gives the same values
    with torch.no_grad():
        out2, _ = layer(x.to(DEV), attention_mask=None, rotary_pos_emb=freqs.to(DEV))
    tol("out2, out", rel_l2(out2, out), 2e-3)


def test_embedding_and_masked_output_layer_modules_against_the_reference_fixtures(megatron):
    """The two classes the reference replaces outright, constructed with Megatron's signatures, against fixtures made by the
    reference's OWN code (oracle/make_golden.py): embedding_scatter.pt (LanguageModelEmbedding.forward, every
    external_feature_dict form) bit-exact; masked_linear_bf16.pt (LinearWithGradAccumulationAndAsyncCommunication forward +
    backward with a logit_mask, bf16) for the HIP ColumnParallelLinear's forward and autograd backward."""
    from conftest import load_golden
    emb_cls = sys.modules["megatron.core.models.common.embeddings.language_model_embedding"].LanguageModelEmbedding
    cpl_cls = sys.modules["megatron.core.tensor_parallel.layers"].ColumnParallelLinear
    g = load_golden("embedding_scatter.pt")
    V, H = g["weight"].shape
    cfg = dm.TransformerConfig(hidden_size=H, params_dtype=torch.float32)                # the fixture's table is fp32: rows move bit for bit
    emb = emb_cls(config=cfg, vocab_size=V, max_sequence_length=g["ids"].shape[1], position_embedding_type="rope")
    emb.load_state_dict({"word_embeddings.weight": g["weight"]})
    ids, feats = g["ids"].to(DEV), g["feats"].to(DEV)
    cases = [("plain", ids, None), ("with_indices", ids, {"features": feats, "indices": g["indices"].to(DEV)}),
             ("with_pre_len", ids[:1].repeat(3, 1), {"features": feats, "pre_len": 5}),
             ("with_src_tgt", ids, {"features": feats, "src_indices": g["src"].to(DEV), "tgt_indices": g["tgt"].to(DEV)})]
    for name, tok, efd in cases:
        with torch.no_grad():
            out = emb(tok, None, external_feature_dict=efd)
        assert torch.equal(out.cpu(), g[name]), name                                      # gather / scatter: bit-exact
    # gradients (bf16, the dtype of the path): the table gets the fp32 scatter-add of the rows that kept their word embedding,
    # the features the gathered rows — against torch autograd over the reference's expression (:102-131)
    emb16 = emb_cls(config=dm.TransformerConfig(hidden_size=H), vocab_size=V, max_sequence_length=g["ids"].shape[1],
                    position_embedding_type="rope")
    emb16.load_state_dict({"word_embeddings.weight": g["weight"].bfloat16()})
    f = feats.bfloat16().requires_grad_(True)
    out = emb16(ids, None, external_feature_dict={"features": f, "src_indices": g["src"].to(DEV), "tgt_indices": g["tgt"].to(DEV)})
    go = torch.randn(out.shape, generator=torch.Generator().manual_seed(1)).bfloat16()
    out.backward(go.to(DEV))
    wr = g["weight"].bfloat16().float().requires_grad_(True)
    fr = g["feats"].bfloat16().float().requires_grad_(True)
    we = wr[g["ids"]].clone()
    we[g["tgt"][0], g["tgt"][1]] = fr[g["src"][0], g["src"][1]]
    we.transpose(0, 1).contiguous().backward(go.float())
    tol("f.grad, fr.grad", rel_l2(f.grad, fr.grad), 1e-6)                      # pure row moves
    tol("emb16.word_embeddings.weight.grad, wr.grad", rel_l2(emb16.word_embeddings.weight.grad, wr.grad), 4e-3)       # fp32 sums of bf16 rows, rounded once to bf16

    m = load_golden("masked_linear_bf16.pt")
    n_out, n_in = m["w"].shape
    lin = cpl_cls(n_in, n_out, config=dm.TransformerConfig(hidden_size=n_in), init_method=None, bias=False, skip_bias_add=False)
    lin.load_state_dict({"weight": m["w"]})
    x = m["x"].to(DEV).requires_grad_(True)
    out, bias = lin(x, logit_mask=m["mask"].to(DEV))
    assert bias is None and tuple(out.shape) == tuple(m["y"].shape)
    e_y = rel_l2(out, m["y"])
    out.backward(m["go"].to(DEV))
    e_dx, e_dw = rel_l2(x.grad, m["dx"]), rel_l2(lin.weight.grad, m["dw"])
    assert max(e_y, e_dx, e_dw) < 4e-3, (e_y, e_dx, e_dw)
    keep = m["mask"][0].to(DEV)
    assert float(x.grad[~keep].abs().max()) == 0.0           # zeros.masked_scatter (M/core/tensor_parallel/layers.py:455-460)


def test_sequence_parallel_tensor_parallel_layer_matches_the_unsharded_oracle(megatron, monkeypatch):
    """`--sequence-parallel` with TP = 2 (every reference script passes it, stage3 .sh:151; BASELINE config 5): the Megatron-built
    layer on two simulated tensor-parallel ranks — column-parallel linears all-gather their sequence shard over the TP group
    (M/core/tensor_parallel/layers.py:392-399) and reduce-scatter the input gradient (:483-494), row-parallel linears
    reduce-scatter their output (:1095) — reassembled == the unsharded oracle layer, forward and backward."""
    from long_vita_amd import parallel_state as mpu, tensor_parallel as tpar
    from long_vita_amd.gpt_vl_model import GPTConfig
    from test_train_gpu import _run_grid
    tp, S = 2, 512
    ocfg = ollm.LLMConfig(**CFG)
    p = ollm.init_llm_params(ocfg, seed=41)
    lp = p["layers"][0]
    g = torch.Generator().manual_seed(42)
    x = (torch.randn(S, 1, CFG["hidden"], generator=g) * 0.5).bfloat16()
    w_out = torch.randn(S, 1, CFG["hidden"], generator=g).bfloat16()
    freqs = glue.rope_emb(S, glue.rope_inv_freq(ocfg.head_dim, ocfg.rope_theta))
    xo = x.clone().requires_grad_(True)
    lpo = {k: v.clone().requires_grad_(True) for k, v in lp.items()}
    ref, _ = ollm.decoder_layer(xo, lpo, ocfg, freqs, lambda q, k, v: oattn.core_attention(q, k, v, causal=True))
    (ref.float() * w_out.float()).sum().backward()

    def rank_fn(ci, ti):
        # the simulated ranks are threads: autograd must run each rank's backward (with its collectives) on that rank's
        # own thread, not on the engine's shared device thread
        with torch.autograd.set_multithreading_enabled(False):
            return rank_body(ci, ti)

    def rank_body(ci, ti):
        shard, _ = tpar.shard_llm_params(p, GPTConfig(**CFG), tp, ti)
        sl = shard["layers"][0]
        mcfg = dm.TransformerConfig(hidden_size=CFG["hidden"], num_attention_heads=CFG["heads"], num_query_groups=CFG["kv_groups"],
                                    kv_channels=CFG["head_dim"], ffn_hidden_size=CFG["ffn"], sequence_parallel=True,
                                    tensor_model_parallel_size=tp)
        layer = dm.build_module(megatron.get_gpt_layer_with_transformer_engine_spec(), config=mcfg, layer_number=1)
        assert layer.self_attention.linear_qkv.sequence_parallel and layer.mlp.linear_fc2.sequence_parallel
        assert tuple(layer.self_attention.linear_qkv.weight.shape) == tuple(sl["qkv_w"].shape)
        _load(layer, sl, True)
        n = S // tp
        xs = x[ti * n:(ti + 1) * n].to(DEV).requires_grad_(True)                     # this rank's sequence shard
        out, _ = layer(xs, attention_mask=None, rotary_pos_emb=freqs.to(DEV))
        assert out.shape == (n, 1, CFG["hidden"])
        (out.float() * w_out[ti * n:(ti + 1) * n].to(DEV).float()).sum().backward()
        grads = {k: v.grad.detach().clone() for k, v in layer.named_parameters()}
        return out.detach(), xs.grad.detach(), grads

    outs = _run_grid(tp, 1, rank_fn, {"mpu": mpu}, monkeypatch)
    out = torch.cat([outs[(0, t)][0] for t in range(tp)], 0)
    dx = torch.cat([outs[(0, t)][1] for t in range(tp)], 0)
    tol("out, ref", rel_l2(out, ref), 4.1e-03)
    tol("dx, xo.grad", rel_l2(dx, xo.grad), 4.7e-03)
    gs = [outs[(0, t)][2] for t in range(tp)]
    d, qpg, ng = CFG["head_dim"], CFG["heads"] // CFG["kv_groups"], CFG["kv_groups"]
    cat0 = lambda k: torch.cat([g_[k] for g_ in gs], 0)                               # noqa: E731
    errs = {"qkv_w": rel_l2(cat0("self_attention.linear_qkv.weight"), lpo["qkv_w"].grad),
            "qkv_b": rel_l2(cat0("self_attention.linear_qkv.bias"), lpo["qkv_b"].grad),
            "o_w": rel_l2(torch.cat([g_["self_attention.linear_proj.weight"] for g_ in gs], 1), lpo["o_w"].grad),
            "fc2_w": rel_l2(torch.cat([g_["mlp.linear_fc2.weight"] for g_ in gs], 1), lpo["fc2_w"].grad),
            # sequence-parallel parameters (norm weights): each rank saw its sequence shard, Megatron all-reduces their grads
            "ln1": rel_l2(sum(g_["self_attention.linear_qkv.layer_norm_weight"].float() for g_ in gs), lpo["ln1"].grad),
            "ln2": rel_l2(sum(g_["mlp.linear_fc1.layer_norm_weight"].float() for g_ in gs), lpo["ln2"].grad)}
    halves = [g_["mlp.linear_fc1.weight"].chunk(2, 0) for g_ in gs]
    errs["fc1_w"] = rel_l2(torch.cat([h_[0] for h_ in halves] + [h_[1] for h_ in halves], 0), lpo["fc1_w"].grad)
    _record("sp_tp2_layer", errs)
    tol("worst gradient", max(errs.values()), 9.4e-03)


@pytest.mark.parametrize("tp,cp", [(1, 2), (2, 2), (1, 4)])
def test_context_parallel_layer_trains_through_the_megatron_built_module(megatron, monkeypatch, tp, cp):
    """VERDICT r2 "missing" 1 / BASELINE config 5 (TP = 2 x CP = 4 training): the Megatron-built TE-spec layer on simulated
    TP x CP ranks with autograd ON.  core_attention = HipDotProductAttention -> autograd_fns.FlashAttnCPFn (K / V all-gather over
    the CP group forward, dK / dV reduce-scatter backward — TE's AttnFuncWithCP behind M/core/models/gpt/gpt_layer_specs.py:40);
    every rank holds its two zig-zag chunks (M/training/utils.py:329-341), under TP = 2 also `--sequence-parallel`.  Reassembled
    output, input gradient and EVERY parameter gradient (summed over the CP ranks, concatenated over the TP ranks) vs the
    unsharded oracle layer and torch autograd over it."""
    from long_vita_amd import parallel_state as mpu, tensor_parallel as tpar, training_utils
    from long_vita_amd.gpt_vl_model import GPTConfig
    from test_train_gpu import _run_grid
    S = 2048 if cp == 4 else 1024
    ocfg = ollm.LLMConfig(**CFG)
    p = ollm.init_llm_params(ocfg, seed=51)
    lp = p["layers"][0]
    g = torch.Generator().manual_seed(52)
    x = (torch.randn(S, 1, CFG["hidden"], generator=g) * 0.5).bfloat16()
    w_out = torch.randn(S, 1, CFG["hidden"], generator=g).bfloat16()
    freqs = glue.rope_emb(S, glue.rope_inv_freq(ocfg.head_dim, ocfg.rope_theta))
    xo = x.clone().requires_grad_(True)
    lpo = {k: v.clone().requires_grad_(True) for k, v in lp.items()}
    ref, _ = ollm.decoder_layer(xo, lpo, ocfg, freqs, lambda q, k, v: oattn.core_attention(q, k, v, causal=True))
    (ref.float() * w_out.float()).sum().backward()

    def rank_fn(ci, ti):
        with torch.autograd.set_multithreading_enabled(False):       # each simulated rank's backward (and its collectives) on its own thread
            return rank_body(ci, ti)

    def rank_body(ci, ti):
        sl = tpar.shard_llm_params(p, GPTConfig(**CFG), tp, ti)[0]["layers"][0] if tp > 1 else lp
        mcfg = dm.TransformerConfig(hidden_size=CFG["hidden"], num_attention_heads=CFG["heads"], num_query_groups=CFG["kv_groups"],
                                    kv_channels=CFG["head_dim"], ffn_hidden_size=CFG["ffn"], sequence_parallel=tp > 1,
                                    tensor_model_parallel_size=tp, context_parallel_size=cp)
        layer = dm.build_module(megatron.get_gpt_layer_with_transformer_engine_spec(), config=mcfg, layer_number=1)
        _load(layer, sl, True)
        zz = lambda t: training_utils.zigzag_slice(t, cp, ci, seq_dim=0)                  # noqa: E731  this CP rank's two chunks
        n = S // cp // tp
        xs = zz(x)[ti * n:(ti + 1) * n].to(DEV).requires_grad_(True)                       # ... and this TP rank's sequence shard
        out, _ = layer(xs, attention_mask=None, rotary_pos_emb=zz(freqs).to(DEV))         # get_pos_emb_on_this_cp_rank (:36-47)
        assert out.shape == (n, 1, CFG["hidden"])
        (out.float() * zz(w_out)[ti * n:(ti + 1) * n].to(DEV).float()).sum().backward()
        return out.detach(), xs.grad.detach(), {k: v.grad.detach().clone() for k, v in layer.named_parameters()}

    outs = _run_grid(tp, cp, rank_fn, {"mpu": mpu}, monkeypatch)

    def unzig(parts):              # parts[ci] = the rank's [2 * C, ...] rows -> global order
        c = S // (2 * cp)
        full = torch.empty((S,) + tuple(parts[0].shape[1:]), dtype=parts[0].dtype, device=parts[0].device)
        for ci, t in enumerate(parts):
            full[ci * c:(ci + 1) * c], full[(2 * cp - 1 - ci) * c:(2 * cp - ci) * c] = t[:c], t[c:]
        return full

    out = unzig([torch.cat([outs[(ci, t)][0] for t in range(tp)], 0) for ci in range(cp)])
    dx = unzig([torch.cat([outs[(ci, t)][1] for t in range(tp)], 0) for ci in range(cp)])
    e_out, e_dx = rel_l2(out, ref), rel_l2(dx, xo.grad)
    # per TP rank: the sum over the CP ranks (Megatron reduces parameter gradients over the DP x CP group)
    gs = [{k: sum(outs[(ci, t)][2][k].float() for ci in range(cp)) for k in outs[(0, t)][2]} for t in range(tp)]
    cat0 = lambda k: torch.cat([g_[k] for g_ in gs], 0)                                  # noqa: E731
    errs = {"out": e_out, "dx": e_dx,
            "qkv_w": rel_l2(cat0("self_attention.linear_qkv.weight"), lpo["qkv_w"].grad),
            "qkv_b": rel_l2(cat0("self_attention.linear_qkv.bias"), lpo["qkv_b"].grad),
            "o_w": rel_l2(torch.cat([g_["self_attention.linear_proj.weight"] for g_ in gs], 1), lpo["o_w"].grad),
            "fc2_w": rel_l2(torch.cat([g_["mlp.linear_fc2.weight"] for g_ in gs], 1), lpo["fc2_w"].grad)}
    # norm weights: replicated at TP = 1; sequence-parallel parameters under TP > 1 (each rank saw its shard, Megatron all-reduces)
    ln = lambda k: (sum(g_[k] for g_ in gs) if tp > 1 else gs[0][k])                      # noqa: E731
    errs["ln1"] = rel_l2(ln("self_attention.linear_qkv.layer_norm_weight"), lpo["ln1"].grad)
    errs["ln2"] = rel_l2(ln("mlp.linear_fc1.layer_norm_weight"), lpo["ln2"].grad)
    halves = [g_["mlp.linear_fc1.weight"].chunk(2, 0) for g_ in gs]
    errs["fc1_w"] = rel_l2(torch.cat([h_[0] for h_ in halves] + [h_[1] for h_ in halves], 0), lpo["fc1_w"].grad)
    _record("cp_layer_tp%d_cp%d" % (tp, cp), errs)
    tol("forward", e_out, 4.0e-03)
    tol("worst gradient", max(errs.values()), 9.7e-03)


def test_embedding_backward_under_sequence_parallelism_reaches_every_vocab_row_and_the_projector(megatron, monkeypatch):
    """with `--sequence-parallel` the embedding output is scattered along the sequence
    (scatter_to_sequence_parallel_region, language_model_embedding.py:157-160) and its BACKWARD is an all-gather: each tensor-parallel
    rank owns a vocabulary slice hit by tokens of every sequence shard, and the replicated projector must receive the full
    feature gradient on every rank.  Two simulated TP ranks vs torch autograd over the reference's expression, unsharded."""
    from long_vita_amd import parallel_state as mpu
    from test_train_gpu import _run_grid
    emb_cls = sys.modules["megatron.core.models.common.embeddings.language_model_embedding"].LanguageModelEmbedding
    tp, V, H, S, N, L = 2, 512, 256, 256, 3, 16
    gen = torch.Generator().manual_seed(77)
    table = torch.randn(V, H, generator=gen).bfloat16()
    ids = torch.randint(0, V, (1, S), generator=gen)
    feats = torch.randn(N, L, H, generator=gen).bfloat16()
    pos = torch.randperm(S, generator=gen)[:N * L].sort().values                     # visual rows spread over BOTH sequence shards
    indices = torch.stack([torch.zeros(N, L, dtype=torch.int64), pos.view(N, L)])
    go = torch.randn(S, 1, H, generator=gen).bfloat16()
    wr, fr = table.float().requires_grad_(True), feats.float().requires_grad_(True)
    we = wr[ids].clone()
    we[indices[0].reshape(-1), indices[1].reshape(-1)] = fr.reshape(-1, H)
    we.transpose(0, 1).contiguous().backward(go.float())

    def rank_fn(ci, ti):
        with torch.autograd.set_multithreading_enabled(False):
            cfg = dm.TransformerConfig(hidden_size=H, sequence_parallel=True, tensor_model_parallel_size=tp)
            emb = emb_cls(config=cfg, vocab_size=V, max_sequence_length=S, position_embedding_type="rope")
            emb.load_state_dict({"word_embeddings.weight": table[ti * V // tp:(ti + 1) * V // tp]})
            f = feats.to(DEV).requires_grad_(True)
            out = emb(ids.to(DEV), None, external_feature_dict={"features": f, "indices": indices.to(DEV)})
            n = S // tp
            assert out.shape == (n, 1, H)
            out.backward(go[ti * n:(ti + 1) * n].to(DEV))
            return out.detach(), f.grad.detach(), emb.word_embeddings.weight.grad.detach()

    outs = _run_grid(tp, 1, rank_fn, {"mpu": mpu}, monkeypatch)
    assert torch.equal(torch.cat([outs[(0, t)][0] for t in range(tp)], 0).cpu(), we.detach().transpose(0, 1).bfloat16())
    for t in range(tp):
        tol("feature gradient on every TP rank", rel_l2(outs[(0, t)][1], fr.grad), 1e-6)               # pure row moves
    tol("vocab-parallel table gradient", rel_l2(torch.cat([outs[(0, t)][2] for t in range(tp)], 0), wr.grad), 1.3e-03)
    hit_other_shard = int((wr.grad[:V // tp].abs().sum(-1) > 0).sum())                # rank 0's rows hit from rank 1's sequence shard
    assert hit_other_shard > 0


@pytest.mark.parametrize("spec", ["local", "te"])
def test_vit_layer_built_by_megatron_matches_the_oracle_forward_and_backward(megatron, spec):
    """VERDICT r2 "missing" 2 + 4: the InternViT layer built by `build_module(get_vit_layer_*_for_intern())` — the builders
    MegatronVisionModel.__init__ takes from M/core/models/vision/vit_layer_specs.py:55-101 (M/pretrain_long_vita.py:337-356), swapped
    by the adaptor for this package's — with autograd ON (reference stage 2 trains the encoder): LayerNorm -> QKV (+ bias) -> non-causal
    attention over 3 frames x 1025 tokens x 16 heads x 64 -> proj -> `residual + (out + bias) * ls1` -> LayerNorm -> fc1 + GELU -> fc2 ->
    `residual + (out + bias) * ls2`.  Output, input gradient and all 14 parameter gradients vs torch autograd over oracle.vit.vit_layer
    (bf16 rounding chain; pinned to the reference's HF InternVisionModel by hf_vit.pt); the no-grad call takes the fused-epilogue path."""
    from oracle import vit as ovit
    vls = sys.modules["long_vita_megatron.core.models.vision.vit_layer_specs"]
    vcfg = ovit.ViTConfig(num_layers=1, unfused_bias=spec == "local")        # local spec: Megatron MLP adds fc1's bias unfused (:213)
    vp = ovit.init_vit_params(vcfg, seed=61)
    gen = torch.Generator().manual_seed(62)
    lp = {k: v.clone() for k, v in vp["layers"][0].items()}
    for k in ("ln1_w", "ln2_w"):
        lp[k] = (1 + 0.1 * torch.randn(lp[k].shape, generator=gen)).bfloat16()
    for k in ("ln1_b", "ln2_b"):
        lp[k] = (0.1 * torch.randn(lp[k].shape, generator=gen)).bfloat16()
    lp["ls1"] = (0.1 + 0.02 * torch.randn(lp["ls1"].shape, generator=gen)).bfloat16()
    lp["ls2"] = (0.1 + 0.02 * torch.randn(lp["ls2"].shape, generator=gen)).bfloat16()
    n, S, H = 3, vcfg.seq, vcfg.hidden
    x = (torch.randn(n, S, H, generator=gen) * 0.5).bfloat16()
    go = torch.randn(n, S, H, generator=gen).bfloat16()
    xo = x.clone().requires_grad_(True)
    lpo = {k: v.clone().requires_grad_(True) for k, v in lp.items()}
    ref = ovit.vit_layer(xo, lpo, vcfg)
    ref.backward(go)

    mcfg = dm.TransformerConfig(hidden_size=H, num_attention_heads=vcfg.heads, num_query_groups=vcfg.heads, kv_channels=vcfg.head_dim,
                                ffn_hidden_size=vcfg.ffn, normalization="LayerNorm", layernorm_epsilon=vcfg.ln_eps, add_bias_linear=True,
                                add_qkv_bias=True, gated_linear_unit=False, activation_func=torch.nn.functional.gelu)
    builder = vls.get_vit_layer_local_spec_for_intern if spec == "local" else vls.get_vit_layer_with_transformer_engine_spec_for_intern
    layer = dm.build_module(builder(), config=mcfg, layer_number=1)
    assert type(layer).__name__ == "InternViTTransformerLayer" and all(q.is_cuda for q in layer.parameters())
    n1 = ("input_layernorm.weight", "input_layernorm.bias") if spec == "local" else (
        "self_attention.linear_qkv.layer_norm_weight", "self_attention.linear_qkv.layer_norm_bias")
    n2 = ("pre_mlp_layernorm.weight", "pre_mlp_layernorm.bias") if spec == "local" else (
        "mlp.linear_fc1.layer_norm_weight", "mlp.linear_fc1.layer_norm_bias")
    names = {"ln1_w": n1[0], "ln1_b": n1[1], "qkv_w": "self_attention.linear_qkv.weight", "qkv_b": "self_attention.linear_qkv.bias",
             "proj_w": "self_attention.linear_proj.weight", "proj_b": "self_attention.linear_proj.bias", "ls1": "ls1",
             "ln2_w": n2[0], "ln2_b": n2[1], "fc1_w": "mlp.linear_fc1.weight", "fc1_b": "mlp.linear_fc1.bias",
             "fc2_w": "mlp.linear_fc2.weight", "fc2_b": "mlp.linear_fc2.bias", "ls2": "ls2"}
    assert set(names.values()) == {k for k, _ in layer.named_parameters()}           # the checkpoint names of the reference's layer
    layer.load_state_dict({v: lp[k].to(DEV) for k, v in names.items()})
    xh = x.transpose(0, 1).contiguous().to(DEV).requires_grad_(True)                  # Megatron's [s, b, h]
    out, _ = layer(xh, attention_mask=None)
    assert out.shape == (S, n, H) and out.dtype == torch.bfloat16
    tol("forward", rel_l2(out.transpose(0, 1), ref), 1.4e-3)                   # measured 9.0e-4
    out.backward(go.transpose(0, 1).contiguous().to(DEV))
    params = dict(layer.named_parameters())
    errs = {"dx": rel_l2(xh.grad.transpose(0, 1), xo.grad)}
    for k, nme in names.items():
        assert params[nme].grad is not None, nme
        errs[k] = rel_l2(params[nme].grad, lpo[k].grad)
    _record("vit_layer_" + spec, errs)
    tol("worst gradient", max(errs.values()), 6.6e-3)                          # measured 4.4e-3 (ls2)
    with torch.no_grad():                                                             # inference: fc1 + bias + GELU in one GEMM epilogue
        out2, _ = layer(x.transpose(0, 1).contiguous().to(DEV), attention_mask=None)
    tol("no-grad path vs autograd path", rel_l2(out2, out), 1e-5)              # measured 0: the same rounding chain


def test_siglip_layer_built_by_megatron_at_its_real_sizes(megatron):
    """BASELINE config 2 names SigLIP next to InternViT: `get_vit_layer_local_spec_for_siglip` (M/core/models/vision/vit_layer_specs.py:30-53,
    taken by MegatronVisionModel for `--vision-model-type siglip_400m`, M/pretrain_long_vita.py:268-307,366-370) at SigLIP-400M's own
    sizes — hidden 1152, 16 heads x 72, FFN 4304, tanh GELU, no LayerScale, 1024 tokens — none of which the MFMA kernels tile: the head
    size is zero-padded to 128 inside HipDotProductAttention, the 4304-deep contractions (fc2 forward, fc1 dgrad) to 4352 inside
    ops.gemm; the biases of proj / fc1 / fc2 meet the bf16-rounded product in ops of their own (skip_bias_add).  Output, input gradient
    and all 12 parameter gradients vs torch autograd over oracle.vit.vit_layer(siglip_400m) (pinned on transformers' SiglipVisionModel
    by siglip_vit.pt); the no-grad call takes the fused BIAS2_GELU_TANH epilogue."""
    import functools
    from oracle import vit as ovit
    vls = sys.modules["long_vita_megatron.core.models.vision.vit_layer_specs"]
    vcfg = ovit.ViTConfig.siglip_400m(num_layers=1)
    vp = ovit.init_vit_params(vcfg, seed=71)
    gen = torch.Generator().manual_seed(72)
    lp = {k: v.clone() for k, v in vp["layers"][0].items() if k not in ("ls1", "ls2")}
    for k in ("ln1_w", "ln2_w"):
        lp[k] = (1 + 0.1 * torch.randn(lp[k].shape, generator=gen)).bfloat16()
    for k in ("ln1_b", "ln2_b", "qkv_b", "proj_b", "fc1_b", "fc2_b"):
        lp[k] = (0.1 * torch.randn(lp[k].shape, generator=gen)).bfloat16()
    n, S, H = 2, vcfg.seq, vcfg.hidden
    assert (S, H, vcfg.head_dim, vcfg.ffn) == (1024, 1152, 72, 4304)
    x = (torch.randn(n, S, H, generator=gen) * 0.5).bfloat16()
    go = torch.randn(n, S, H, generator=gen).bfloat16()
    xo = x.clone().requires_grad_(True)
    lpo = {k: v.clone().requires_grad_(True) for k, v in lp.items()}
    ref = ovit.vit_layer(xo, lpo, vcfg)
    ref.backward(go)

    mcfg = dm.TransformerConfig(hidden_size=H, num_attention_heads=vcfg.heads, num_query_groups=vcfg.heads, kv_channels=vcfg.head_dim,
                                ffn_hidden_size=vcfg.ffn, normalization="LayerNorm", layernorm_epsilon=vcfg.ln_eps, add_bias_linear=True,
                                add_qkv_bias=True, gated_linear_unit=False,
                                activation_func=functools.partial(torch.nn.functional.gelu, approximate="tanh"))
    layer = dm.build_module(vls.get_vit_layer_local_spec_for_siglip(), config=mcfg, layer_number=1)
    assert type(layer).__name__ == "SigLIPViTTransformerLayer" and all(q.is_cuda for q in layer.parameters())
    names = {"ln1_w": "input_layernorm.weight", "ln1_b": "input_layernorm.bias", "qkv_w": "self_attention.linear_qkv.weight",
             "qkv_b": "self_attention.linear_qkv.bias", "proj_w": "self_attention.linear_proj.weight",
             "proj_b": "self_attention.linear_proj.bias", "ln2_w": "pre_mlp_layernorm.weight", "ln2_b": "pre_mlp_layernorm.bias",
             "fc1_w": "mlp.linear_fc1.weight", "fc1_b": "mlp.linear_fc1.bias", "fc2_w": "mlp.linear_fc2.weight", "fc2_b": "mlp.linear_fc2.bias"}
    assert set(names.values()) == {k for k, _ in layer.named_parameters()}           # no ls1 / ls2: siglip_vit_model.py has no LayerScale
    assert tuple(dict(layer.named_parameters())["mlp.linear_fc2.weight"].shape) == (1152, 4304)     # Megatron's shapes, not padded ones
    layer.load_state_dict({v: lp[k].to(DEV) for k, v in names.items()})
    xh = x.transpose(0, 1).contiguous().to(DEV).requires_grad_(True)
    out, _ = layer(xh, attention_mask=None)
    assert out.shape == (S, n, H) and out.dtype == torch.bfloat16
    e_fwd = rel_l2(out.transpose(0, 1), ref)
    out.backward(go.transpose(0, 1).contiguous().to(DEV))
    params = dict(layer.named_parameters())
    errs = {"out": e_fwd, "dx": rel_l2(xh.grad.transpose(0, 1), xo.grad)}
    for k, nme in names.items():
        assert params[nme].grad is not None, nme
        errs[k] = rel_l2(params[nme].grad, lpo[k].grad)
    _record("siglip_layer_local", errs)
    tol("forward", e_fwd, 4.4e-3)                                                    # measured 2.9e-3 (no LayerScale: the branch outputs enter the stream at full size)
    tol("worst gradient", max(v for k, v in errs.items() if k != "out"), 7.4e-3)     # measured 4.9e-3 (ln1_w)
    with torch.no_grad():
        out2, _ = layer(x.transpose(0, 1).contiguous().to(DEV), attention_mask=None)
    tol("no-grad path vs autograd path", rel_l2(out2, out), 1e-5)                    # measured 0: the same rounding chain


# ---------------------------------------------------------------------------------------------------------------------------------
# r04 — the rest of the path under the reference's entry classes (VERDICT r3 "missing" 1): ViT front end, downsample + projector,
# final norm, loss; activation recompute re-entering the autograd Functions
# ---------------------------------------------------------------------------------------------------------------------------------
def _vision_model(ovit, vcfg, vp, args_over=None, vit_grad=True, llm_cfg=None, gpt_kwargs=None):
    """MegatronVisionModel (the entry script's class: tests/dummy_megatron.py restates it in plain torch) constructed through the
    PATCHED names — InternViTModel, the ViT layer spec, MultimodalProjector — inside a GPTVLModel whose __init__ the adaptor wrapped."""
    import types
    H = vcfg.hidden
    vit_cfg = dm.TransformerConfig(num_layers=vcfg.num_layers, hidden_size=H, num_attention_heads=vcfg.heads, num_query_groups=vcfg.heads,
                                   kv_channels=vcfg.head_dim, ffn_hidden_size=vcfg.ffn, normalization="LayerNorm",
                                   layernorm_epsilon=vcfg.ln_eps, add_bias_linear=True, add_qkv_bias=True, gated_linear_unit=False,
                                   activation_func=torch.nn.functional.gelu)
    proj_cfg = dm.TransformerConfig(hidden_size=vcfg.llm_hidden, ffn_hidden_size=H, add_bias_linear=False, gated_linear_unit=False,
                                    activation_func=torch.nn.functional.gelu)              # M/pretrain_long_vita.py:397-410
    a = dict(vision_seq_length=vcfg.seq, image_token_length=256, vision_model_type="intern_300m", vision_context_parallel=False,
             vision_downsample_ratio=0.5, vision_downsample_stride=1, add_class_token=True, vision_model_freeze=not vit_grad,
             vision_projector_freeze=False, vision_model_recompute=False, vision_projector_recompute=False, patch_dim=14,
             image_size=vcfg.image, vision_projector_pre_norm=True)
    a.update(args_over or {})
    args = types.SimpleNamespace(**a)
    gpt_cls = sys.modules["long_vita_megatron.core.models.multimodal.gpt_vl_model"].GPTVLModel
    model = gpt_cls(llm_cfg or proj_cfg, external_feature_model_provider=lambda cfg: dm.MegatronVisionModel(args, vit_cfg, proj_cfg).to(DEV).bfloat16(),
                    **(gpt_kwargs or {}))                                            # Float16Module
    efm = model.external_feature_model
    sd = {"vit.conv1.weight": vp["conv_w"], "vit.conv1.bias": vp["conv_b"], "vit.class_token": vp["cls"],
          "vit.position_embeddings.weight": vp["pos"], "pre_proj_layernorm.weight": vp["proj_ln_w"],
          "pre_proj_layernorm.bias": vp["proj_ln_b"], "vision_projection.encoder.linear_fc1.weight": vp["proj_fc1"],
          "vision_projection.encoder.linear_fc2.weight": vp["proj_fc2"]}
    lnames = {"ln1_w": "input_layernorm.weight", "ln1_b": "input_layernorm.bias", "qkv_w": "self_attention.linear_qkv.weight",
              "qkv_b": "self_attention.linear_qkv.bias", "proj_w": "self_attention.linear_proj.weight",
              "proj_b": "self_attention.linear_proj.bias", "ls1": "ls1", "ln2_w": "pre_mlp_layernorm.weight",
              "ln2_b": "pre_mlp_layernorm.bias", "fc1_w": "mlp.linear_fc1.weight", "fc1_b": "mlp.linear_fc1.bias",
              "fc2_w": "mlp.linear_fc2.weight", "fc2_b": "mlp.linear_fc2.bias", "ls2": "ls2"}
    for i, lp in enumerate(vp["layers"]):
        for k, nme in lnames.items():
            sd[f"vit.decoder.layers.{i}.{nme}"] = lp[k]
    assert set(sd) == {k for k, _ in efm.named_parameters()}, sorted(set(sd) ^ {k for k, _ in efm.named_parameters()})
    efm.load_state_dict({k: v.to(DEV) for k, v in sd.items()})                        # the reference's checkpoint names
    if not vit_grad:
        for k, q in efm.named_parameters():
            if k.startswith("vit."):
                q.requires_grad = False                                               # GPTVLModel.vision_model_freeze (:186-195)
    if gpt_kwargs:
        return efm, sd, model
    return efm, sd


def _randomised_vit_params(ovit, vcfg, seed):
    vp = ovit.init_vit_params(vcfg, seed=seed)
    gen = torch.Generator().manual_seed(seed + 1)
    for lp in vp["layers"]:
        for k in ("ln1_w", "ln2_w"):
            lp[k] = (1 + 0.1 * torch.randn(lp[k].shape, generator=gen)).bfloat16()
        for k in ("ln1_b", "ln2_b"):
            lp[k] = (0.1 * torch.randn(lp[k].shape, generator=gen)).bfloat16()
    vp["proj_ln_w"] = (1 + 0.1 * torch.randn(vp["proj_ln_w"].shape, generator=gen)).bfloat16()
    vp["proj_ln_b"] = (0.1 * torch.randn(vp["proj_ln_b"].shape, generator=gen)).bfloat16()
    vp["cls"] = (0.5 * torch.randn(vp["cls"].shape, generator=gen)).bfloat16()
    return vp


@pytest.mark.parametrize("mode", ["stage2_vit_trains_projector_recompute", "stage3_vit_frozen"])
def test_vision_tower_under_the_reference_entry_classes_forward_and_every_gradient(megatron, mode):
    """`external_feature_model(images=...)` as GPTVLModel.forward calls it (gpt_vl_model.py:285-300): MegatronVisionModel ->
    InternViTModel (conv1 + class token + position embedding + 2 layers through TransformerBlock) -> drop class token + pixel shuffle +
    LayerNorm(4096) -> MultimodalProjector, every class obtained from the dotted name the reference imports it from after the adaptor
    ran.  The torch pieces the reference would run (torch.nn.LayerNorm.forward, the torch pixel_shuffle, Conv2d.forward) are made to
    raise: the library must have displaced them.  Output and EVERY parameter gradient (conv weight / bias, class token, position
    table, 2 x 14 layer parameters, pre-norm weight / bias, both projector weights) vs torch autograd over oracle.vit.vision_model
    (the bf16 chain; pinned on the reference's HF InternVisionModel + ResamplerProjector by hf_vit.pt)."""
    from oracle import vit as ovit
    stage2 = mode.startswith("stage2")
    vcfg = ovit.ViTConfig(num_layers=2, llm_hidden=1024, unfused_bias=True)            # Megatron MLP: bias_activation_fusion off (:213)
    vp = _randomised_vit_params(ovit, vcfg, seed=81)
    gen = torch.Generator().manual_seed(83)
    n = 3
    images = torch.randn(n, 3, vcfg.image, vcfg.image, generator=gen).bfloat16()
    go = torch.randn(n, 256, vcfg.llm_hidden, generator=gen).bfloat16()
    vpo = {k: (v.clone().requires_grad_(True) if torch.is_tensor(v) else [{kk: vv.clone().requires_grad_(True) for kk, vv in lp.items()} for lp in v])
           for k, v in vp.items()}
    ref = ovit.vision_model(images, vpo, vcfg)
    ref.backward(go)

    # stage 2 trains the encoder (its output carries a graph, so tensor_parallel.checkpoint around downsample + projection has an input that
    # requires grad); stage 3 / 4 freeze it (`--vision-model-freeze`, stage3 .sh:203) and run the projector without recompute
    efm, sd = _vision_model(ovit, vcfg, vp, dict(vision_projector_recompute=stage2), vit_grad=stage2)
    assert getattr(efm, "_vita_hip_installed", False) and type(efm.vit).__module__ == "long_vita_amd.vision_modules"
    assert type(efm.vision_projection).__module__ == "long_vita_amd.vision_modules"

    def boom(*a, **k):
        raise AssertionError("a torch op of the reference's vision path ran")
    efm.pre_proj_layernorm.forward = boom
    efm.pixel_shuffle = boom
    efm.vit.conv1.forward = boom
    efm.vit.position_embeddings.forward = boom
    efm.train()
    out = efm(images=images.to(DEV))
    assert out.shape == (n, 256, vcfg.llm_hidden) and out.dtype == torch.bfloat16
    tol("forward", rel_l2(out, ref), 4.9e-3)                                         # measured 3.2e-3
    out.backward(go.to(DEV))
    flat = {"vit.conv1.weight": vpo["conv_w"], "vit.conv1.bias": vpo["conv_b"], "vit.class_token": vpo["cls"],
            "vit.position_embeddings.weight": vpo["pos"], "pre_proj_layernorm.weight": vpo["proj_ln_w"],
            "pre_proj_layernorm.bias": vpo["proj_ln_b"], "vision_projection.encoder.linear_fc1.weight": vpo["proj_fc1"],
            "vision_projection.encoder.linear_fc2.weight": vpo["proj_fc2"]}
    okeys = {"input_layernorm.weight": "ln1_w", "input_layernorm.bias": "ln1_b", "self_attention.linear_qkv.weight": "qkv_w",
             "self_attention.linear_qkv.bias": "qkv_b", "self_attention.linear_proj.weight": "proj_w",
             "self_attention.linear_proj.bias": "proj_b", "ls1": "ls1", "pre_mlp_layernorm.weight": "ln2_w",
             "pre_mlp_layernorm.bias": "ln2_b", "mlp.linear_fc1.weight": "fc1_w", "mlp.linear_fc1.bias": "fc1_b",
             "mlp.linear_fc2.weight": "fc2_w", "mlp.linear_fc2.bias": "fc2_b", "ls2": "ls2"}
    errs = {}
    for name, q in efm.named_parameters():
        if name.startswith("vit.") and not stage2:
            assert q.grad is None, name                                                # frozen encoder: nothing reaches it
            continue
        if name in flat:
            want = flat[name].grad
        else:
            _, _, _, li, rest = name.split(".", 4)
            want = vpo["layers"][int(li)][okeys[rest]].grad
        assert q.grad is not None, name
        errs[name] = rel_l2(q.grad, want)
    _record("vision_tower_" + mode, dict(errs, out=rel_l2(out, ref)))
    assert len(errs) == (36 if stage2 else 4)
    tol("worst gradient", max(errs.values()), 9.8e-3 if stage2 else 5.6e-3)              # measured 6.5e-3 (class token) / 3.7e-3
    with torch.no_grad():                                                              # inference: the fused epilogues, no autograd nodes
        efm.eval()
        out2 = efm(images=images.to(DEV))
    tol("no-grad path vs autograd path", rel_l2(out2, out), 1e-5)                      # measured 0: the same rounding chain


def test_registered_intern_vit_front_end_against_the_references_own_forward(megatron, monkeypatch):
    """`InternViTModel.forward` of the reference (M/core/models/vision/intern_vit_model.py:190-261, imported and run
    on CPU in fp32 over a plain-torch block, oracle/make_golden_composites.py -> intern_vit_forward.pt) against the class this package
    registers on the same dotted name, built with the reference's constructor call and loaded with the same state dict: the input the
    block receives (conv1 as patchify + GEMM, class token, position rows, [s, b, h]), the output after the same plain-torch block, and
    the gradients of conv1 / class token / position table.  With and without the class token (position ids 1.. of a table one row longer,
    class_token kept but frozen)."""
    from conftest import load_golden
    from oracle import leaves, make_golden_composites as comp
    g = load_golden("intern_vit_forward.pt")
    monkeypatch.setattr(sys.modules["megatron.core.transformer.transformer_block"], "TransformerBlock", leaves.Block)
    cls = sys.modules["long_vita_megatron.core.models.vision.intern_vit_model"].InternViTModel
    for c in g["cases"]:
        cfg = dm.TransformerConfig(hidden_size=c["hidden"], params_dtype=torch.bfloat16)
        vit = cls(cfg, "spec", add_class_token=c["add_class_token"], patch_dim=c["patch"], img_h=c["img"], img_w=c["img"])
        assert sorted(vit.state_dict()) == c["state_keys"] and [n for n, _ in vit.named_parameters()] == c["param_names"]
        assert vit.seq_length == c["seq_length"] and torch.equal(vit.position_ids.cpu(), c["position_ids"])
        assert bool(vit.class_token.requires_grad) == c["class_token_requires_grad"]
        vit.decoder = vit.decoder.to(DEV)
        leaves.init_by_name(vit, seed=11)                                              # the fixture's weights, by parameter name
        vit = vit.to(torch.bfloat16)
        seen = {}
        h = vit.decoder.register_forward_hook(lambda m, a, o: seen.update(x=a[0].detach().clone()))      # (returns None: the output stays)
        x = comp.vit_case_inputs(c).bfloat16().to(DEV)
        out = vit(x)
        h.remove()
        assert out.shape == c["out"].shape and seen["x"].shape == c["block_input"].shape
        tol(f"block input ({c['name']})", rel_l2(seen["x"], c["block_input"]), 6e-3)      # bf16 weights / images / output vs fp32
        tol(f"output ({c['name']})", rel_l2(out, c["out"]), 8e-3)
        w = torch.linspace(-1, 1, out.numel()).view_as(out).to(DEV)
        (out.float() * w).sum().backward()
        for n, p in vit.named_parameters():
            want = c["grads"][n]
            if want is None:
                assert p.grad is None, n
            elif not n.startswith("decoder."):
                # bf16 images / weights / activations against the fp32 fixture, through a saturating tanh block whose input gradient
                # cancels heavily in the column sums: measured 3.9e-2 on d conv1.bias (the worst), limit 1.5 x
                tol(f"d {n} ({c['name']})", rel_l2(p.grad, want), 6e-2)


def test_transformer_block_final_norm_recompute_and_loss_run_on_the_library(megatron):
    """(1) `TransformerBlock(config, spec, post_process=True)` builds `final_layernorm` from the module-level name TENorm
    (M/core/transformer/transformer_block.py:201): after the adaptor that name is layers.Norm -> RMSNorm with library forward / backward.
    (2) `--recompute-granularity full --recompute-method block --recompute-num-layers 1`: Megatron's tensor_parallel.checkpoint re-enters
    this package's autograd Functions in the backward (forward under no_grad, re-run with grad) — outputs and every gradient must
    equal the run without recompute BIT FOR BIT (same kernels, same order).
    (3) `tensor_parallel.vocab_parallel_cross_entropy(logits.float(), labels)` (compute_language_model_loss, gpt_vl_model.py:414):
    loss and d logits vs torch's cross entropy in fp32."""
    from long_vita_amd import layers
    tb_mod = sys.modules["megatron.core.transformer.transformer_block"]
    S = 512
    def build(recompute):
        mcfg = dm.TransformerConfig(num_layers=2, hidden_size=CFG["hidden"], num_attention_heads=CFG["heads"],
                                    num_query_groups=CFG["kv_groups"], kv_channels=CFG["head_dim"], ffn_hidden_size=CFG["ffn"],
                                    recompute_granularity="full" if recompute else None, recompute_method="block" if recompute else None,
                                    recompute_num_layers=1 if recompute else None)
        torch.manual_seed(7)
        blk = tb_mod.TransformerBlock(mcfg, megatron.get_gpt_layer_with_transformer_engine_spec(), post_process=True)
        return blk
    ref_blk = build(False)
    assert isinstance(ref_blk.final_layernorm, layers.RMSNorm)
    blk = build(True)
    blk.load_state_dict(ref_blk.state_dict())
    with torch.no_grad():
        ref_blk.final_layernorm.weight.copy_(1 + 0.1 * torch.randn(CFG["hidden"], generator=torch.Generator().manual_seed(3)).to(DEV))
        blk.final_layernorm.weight.copy_(ref_blk.final_layernorm.weight)
    g = torch.Generator().manual_seed(91)
    x = (torch.randn(S, 1, CFG["hidden"], generator=g) * 0.5).bfloat16().to(DEV)
    go = torch.randn(S, 1, CFG["hidden"], generator=g).bfloat16().to(DEV)
    freqs = glue.rope_emb(S, glue.rope_inv_freq(CFG["head_dim"], 1e6)).to(DEV)
    outs = []
    for b_ in (ref_blk, blk):
        b_.train()
        xi = x.clone().requires_grad_(True)
        o = b_(xi, attention_mask=None, rotary_pos_emb=freqs)
        o.backward(go)
        outs.append((o.detach(), xi.grad, {k: v.grad for k, v in b_.named_parameters()}))
    assert torch.equal(outs[0][0], outs[1][0]) and torch.equal(outs[0][1], outs[1][1])
    for k in outs[0][2]:
        assert outs[1][2][k] is not None, k
        if "norm" in k:                            # fp32 atomic sums over row blocks (arrival order): equal to one bf16 step, as in the tests below
            torch.testing.assert_close(outs[0][2][k].float(), outs[1][2][k].float(), rtol=2 ** -7, atol=1e-6, msg=k)
        else:
            assert torch.equal(outs[0][2][k], outs[1][2][k]), k
    # the final norm against the oracle restatement of RMSNorm on the layer stack's output
    with torch.no_grad():
        pre = ref_blk.final_layernorm
        h = x.clone()
        for layer in ref_blk.layers:
            h, _ = layer(h, attention_mask=None, rotary_pos_emb=freqs)
        want = glue.rmsnorm(h.cpu(), pre.weight.cpu(), 1e-6)
    tol("final_layernorm vs oracle RMSNorm", rel_l2(outs[0][0], want), 1e-4)          # measured 0

    # ---- loss ----------------------------------------------------------------------------------------------------------------------
    tpm = sys.modules["megatron.core.tensor_parallel"]
    n, V = 37, 1536
    logits = (torch.randn(n, 1, V, generator=g) * 3).bfloat16()
    labels = torch.randint(0, V, (n, 1), generator=g)
    gl = torch.rand(n, 1, generator=g)
    lo = logits.float().clone().requires_grad_(True)
    want = torch.nn.functional.cross_entropy(lo.view(n, V), labels.view(n), reduction="none").view(n, 1)
    (want * gl).sum().backward()
    lh = logits.to(DEV).requires_grad_(True)
    loss = tpm.vocab_parallel_cross_entropy(lh.float(), labels.to(DEV))                # as compute_language_model_loss calls it
    assert loss.shape == (n, 1) and loss.dtype == torch.float32
    (loss * gl.to(DEV)).sum().backward()
    tol("loss", rel_l2(loss, want), 1e-5)
    tol("d logits", rel_l2(lh.grad, lo.grad), 2.2e-3)                                    # bf16 leaf: the fp32 gradient is rounded once


def _megatron_vocab_parallel_cross_entropy(shards, target):
    """megatron/core/tensor_parallel/cross_entropy.py (megatron-core, not vendored under /root/reference; the reference calls it through
    LanguageModule.compute_language_model_loss, M/core/models/multimodal/gpt_vl_model.py:414) restated on a list of fp32 vocabulary
    shards, collective by collective: logits_max (all-reduce MAX) -> shard - max -> target_mask = target outside the shard,
    masked_target, predicted_logits[target_mask] = 0 (all-reduce SUM) -> sum_exp (all-reduce SUM) -> loss = log(sum_exp) - predicted;
    backward: softmax with `1 - target_mask` subtracted at masked_target, times the incoming gradient.  Returns loss [n] and
    d loss / d shard for a unit incoming gradient (multiply by g outside)."""
    v_l = shards[0].shape[-1]
    mx = torch.stack([sh.max(dim=-1)[0] for sh in shards]).max(dim=0)[0]
    preds, sums, exps, masks, mts = [], [], [], [], []
    for r, sh in enumerate(shards):
        z = sh - mx.unsqueeze(-1)
        lo, hi = r * v_l, (r + 1) * v_l
        mask = (target < lo) | (target >= hi)
        mt = target.clone() - lo
        mt[mask] = 0
        pred = z[torch.arange(z.shape[0]), mt].clone()
        pred[mask] = 0.0
        e = z.exp()
        preds.append(pred); sums.append(e.sum(-1)); exps.append(e); masks.append(mask); mts.append(mt)
    pred, se = sum(preds), sum(sums)
    loss = se.log() - pred
    grads = []
    for e, mask, mt in zip(exps, masks, mts):
        gsh = e / se.unsqueeze(-1)
        gsh[torch.arange(gsh.shape[0]), mt] -= 1.0 - mask.float()
        grads.append(gsh)
    return loss, grads


@pytest.mark.parametrize("tp", [1, 2])
def test_vocab_parallel_cross_entropy_masks_ignored_targets_and_stays_sharded(megatron, monkeypatch, tp):
    """ADVICE r4 (high + medium).  The stage 1-3 scripts send the datasets' IGNORE_TOKEN_ID = -100 padding labels straight into
    vocab_parallel_cross_entropy (M/pretrain_long_vita.py:751; only stage 4 passes --logit-mask) and rely on Megatron masking them;
    the patched function must give Megatron's loss / gradient for such rows (no exception, no one-hot term) and, under TP = 2, work on
    its vocabulary shard alone: the gradient it returns and every tensor it keeps for the backward have the LOCAL vocabulary width."""
    from test_train_gpu import _run_grid
    from long_vita_amd import parallel_state as mpu
    tpm = sys.modules["megatron.core.tensor_parallel"]
    g = torch.Generator().manual_seed(17)
    n, V = 53, 2048
    v_l = V // tp
    logits = (torch.randn(n, 1, V, generator=g) * 3)
    target = torch.randint(0, V, (n, 1), generator=g)
    target[[0, 7, 20], 0] = -100                                  # padding
    target[31, 0] = V                                             # one past the end
    target[[3, 4], 0] = torch.tensor([v_l - 1, v_l % V])          # both sides of the shard boundary
    gl = torch.rand(n, 1, generator=g)
    shards = [logits[:, 0, r * v_l:(r + 1) * v_l].contiguous() for r in range(tp)]
    want_loss, want_g = _megatron_vocab_parallel_cross_entropy(shards, target[:, 0])
    for i in (0, 7, 20, 31):                                      # the masked rows: log sum exp(l - max), i.e. NOT a skipped row
        assert abs(float(want_loss[i]) - float((logits[i, 0] - logits[i, 0].max()).exp().sum().log())) < 1e-5

    def rank_fn(ci, ti):
        lh = shards[ti].view(n, 1, v_l).to(DEV).requires_grad_(True)
        loss = tpm.vocab_parallel_cross_entropy(lh, target.to(DEV))
        node = loss.grad_fn if "VocabParallel" in type(loss.grad_fn).__name__ else loss.grad_fn.next_functions[0][0]
        kept = [tuple(t.shape) for t in node.saved_tensors]
        (loss * gl.to(DEV)).sum().backward()
        return loss.detach(), lh.grad, kept

    outs = _run_grid(tp, 1, rank_fn, {"mpu": mpu}, monkeypatch)
    for ti in range(tp):
        loss, grad, kept = outs[(0, ti)]
        assert loss.shape == (n, 1) and loss.dtype == torch.float32 and grad.shape == (n, 1, v_l)
        assert all(V not in shp or tp == 1 for shp in kept), kept                  # nothing of the full vocabulary width is kept
        tol(f"loss, TP = {tp}", rel_l2(loss.view(n), want_loss), 1e-5)
        tol(f"d logits shard, TP = {tp}", rel_l2(grad.view(n, v_l), want_g[ti] * gl), 1e-5)
        masked = grad.view(n, v_l)[[0, 7, 20, 31]].cpu()
        assert float(masked.min()) >= 0.0                                           # softmax * g only: no -1 anywhere
    # bf16 logits (the stand-alone step's form) through the same entry point at TP = 1
    if tp == 1:
        lb = logits.bfloat16().to(DEV).requires_grad_(True)
        lossb = tpm.vocab_parallel_cross_entropy(lb, target.to(DEV))
        wl, wg = _megatron_vocab_parallel_cross_entropy([logits.bfloat16().float()[:, 0]], target[:, 0])
        (lossb * gl.to(DEV)).sum().backward()
        tol("loss, bf16 logits", rel_l2(lossb.view(n), wl), 1e-5)
        tol("d logits, bf16 logits", rel_l2(lb.grad.view(n, V), wg[0] * gl), 4e-3)


def test_recompute_keeping_the_attention_result_is_bit_identical_and_skips_the_second_forward(megatron, monkeypatch):
    """`--recompute-granularity full --recompute-method block` through the (patched) tensor_parallel.checkpoint with
    VITA_KEEP_ATTENTION=1 — the checkpointed layers' first run leaves (context, lse) with recompute_cache, the replay in the backward hands
    them to FlashAttnFn instead of launching the forward kernel again.  Output, input gradient, every matrix gradient and the bias
    gradient are BIT identical to the same block without the switch (the norm-weight gradients — fp32 atomic sums — to one bf16 step), and
    the attention forward runs once per layer instead of twice."""
    from long_vita_amd import ops as ops_mod
    tb_mod = sys.modules["megatron.core.transformer.transformer_block"]
    S = 512
    mcfg = dm.TransformerConfig(num_layers=2, hidden_size=CFG["hidden"], num_attention_heads=CFG["heads"], num_query_groups=CFG["kv_groups"],
                                kv_channels=CFG["head_dim"], ffn_hidden_size=CFG["ffn"], recompute_granularity="full",
                                recompute_method="block", recompute_num_layers=2)
    torch.manual_seed(11)
    blk = tb_mod.TransformerBlock(mcfg, megatron.get_gpt_layer_with_transformer_engine_spec(), post_process=True)
    blk.train()
    g = torch.Generator().manual_seed(92)
    x = (torch.randn(S, 1, CFG["hidden"], generator=g) * 0.5).bfloat16().to(DEV)
    go = torch.randn(S, 1, CFG["hidden"], generator=g).bfloat16().to(DEV)
    freqs = glue.rope_emb(S, glue.rope_inv_freq(CFG["head_dim"], 1e6)).to(DEV)
    calls = {"n": 0}
    real = ops_mod.flash_attn

    def counting(*a, **k):
        calls["n"] += 1
        return real(*a, **k)

    monkeypatch.setattr(ops_mod, "flash_attn", counting)
    outs = []
    for keep in ("0", "1"):
        monkeypatch.setenv("VITA_KEEP_ATTENTION", keep)
        calls["n"] = 0
        blk.zero_grad(set_to_none=True)
        xi = x.clone().requires_grad_(True)
        o = blk(xi, attention_mask=None, rotary_pos_emb=freqs)
        o.backward(go)
        outs.append((o.detach().clone(), xi.grad.clone(), {k: v.grad.clone() for k, v in blk.named_parameters()}, calls["n"]))
    assert outs[0][3] == 4 and outs[1][3] == 2, (outs[0][3], outs[1][3])            # 2 layers: forward + recompute vs forward only
    assert torch.equal(outs[0][0], outs[1][0]) and torch.equal(outs[0][1], outs[1][1])
    for k in outs[0][2]:
        if "norm" in k:
            # the norm-weight gradients are fp32 sums over row blocks added with atomics, in ARRIVAL order (bwd.hip rmsnorm_bwd): the two runs
            # compute the same partials and may add them in a different order — equal to the last fp32 bits, i.e. within one bf16 step
            torch.testing.assert_close(outs[0][2][k].float(), outs[1][2][k].float(), rtol=2 ** -7, atol=1e-6, msg=k)
        else:
            assert torch.equal(outs[0][2][k], outs[1][2][k]), k           # GEMM products, and the bias gradient (ordered column sum, ABI 18)


def test_kept_attention_stays_with_its_call_when_a_replayed_call_needs_no_gradient(megatron, monkeypatch):
    """two attention calls inside one checkpointed region, the first on inputs that need no gradient in the replay
    (a frozen layer under an input without grad).  The replayed first call must consume ITS slot (and returns the kept context), so the
    second call's backward sees its own (context, lse): gradients bit-identical to the run without the switch."""
    from long_vita_amd import recompute_cache as rc
    from long_vita_amd.dot_product_attention import HipDotProductAttention
    mcfg = dm.TransformerConfig(num_layers=1, hidden_size=CFG["hidden"], num_attention_heads=CFG["heads"], num_query_groups=CFG["kv_groups"],
                                kv_channels=CFG["head_dim"], ffn_hidden_size=CFG["ffn"])
    att = HipDotProductAttention(mcfg, 1, "causal")
    S, H, G, D = 512, CFG["heads"], CFG["kv_groups"], CFG["head_dim"]
    g = torch.Generator().manual_seed(5)
    mk = lambda h: torch.randn(S, 1, h, D, generator=g).bfloat16().to(DEV)             # noqa: E731
    q0, k0, v0, q1, k1, v1 = mk(H), mk(G), mk(G), mk(H), mk(G), mk(G)
    go = torch.randn(S, 1, H * D, generator=g).bfloat16().to(DEV)

    def region(x):                                   # x carries the gradient; call 0's inputs are constants
        a = att(q0, k0, v0)
        b = att(q1 * x, k1, v1)
        return a + b

    res = []
    for keep in ("0", "1"):
        monkeypatch.setenv("VITA_KEEP_ATTENTION", keep)
        x = torch.ones(1, device=DEV, dtype=torch.bfloat16).requires_grad_(True)
        y = rc.checkpoint_wrapper(dm.checkpoint)(region, False, x)
        y.backward(go)
        res.append((y.detach().clone(), x.grad.clone()))
    assert torch.equal(res[0][0], res[1][0]) and torch.equal(res[0][1], res[1][1])


def test_output_layer_with_a_logit_mask_that_selects_nothing(megatron):
    """under CP the answer tokens of a 128K row all sit on CP rank 0 — every other rank's `logit_mask` is all
    False.  ColumnParallelLinear(output_layer).forward must return an empty [0, b, V] tensor (the reference's masked_select does), and
    its backward zero parameter gradients and a zero input gradient of the full [s, b, c] shape."""
    cpl_cls = sys.modules["megatron.core.tensor_parallel.layers"].ColumnParallelLinear
    cfg = dm.TransformerConfig(hidden_size=256)
    lin = cpl_cls(256, 512, config=cfg, init_method=cfg.init_method, bias=True, skip_bias_add=False)
    x = torch.randn(64, 1, 256, device=DEV).bfloat16().requires_grad_(True)
    mask = torch.zeros(1, 64, dtype=torch.bool, device=DEV)
    out, _ = lin(x, logit_mask=mask)
    assert out.shape == (0, 1, 512)
    out.sum().backward()
    assert x.grad.shape == x.shape and float(x.grad.abs().sum()) == 0.0
    assert float(lin.weight.grad.abs().sum()) == 0.0 and float(lin.bias.grad.abs().sum()) == 0.0


def test_functions_wrapped_in_roctx_ranges_compute_the_same_under_vita_debug():
    """VITA_DEBUG=1 wraps every autograd Function's forward / backward in a roctx range (long_vita_amd/tracing.py): a linear + RMSNorm +
    SwiGLU chain gives bit-identical outputs and gradients with and without the wrappers (run in child processes: the switch is read at
    import)."""
    import os, subprocess, textwrap
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    code = textwrap.dedent("""
        import torch, hashlib
        from long_vita_amd import autograd_fns as F, tracing
        torch.manual_seed(3)
        x = (torch.randn(256, 1, 512, device="cuda") * 0.5).bfloat16().requires_grad_(True)      # [s, b, c]
        w = (torch.randn(1024, 512, device="cuda") * 0.05).bfloat16().requires_grad_(True)
        g = torch.ones(512, device="cuda", dtype=torch.bfloat16)           # (its gradient is an atomic sum: not hashed)
        with tracing.range("test chain"):
            y = F.SwiGLUFn.apply(F.LinearFn.apply(F.RMSNormFn.apply(x, g, 1e-6), w, None, False, False, None))
            y.float().square().sum().backward()
        torch.cuda.synchronize()
        h = hashlib.sha256()
        for t in (y, x.grad, w.grad):
            h.update(t.detach().float().cpu().numpy().tobytes())
        print("WRAPPED" if hasattr(F.LinearFn.forward, "__wrapped__") else "PLAIN", h.hexdigest())
    """)
    outs = {}
    for flag in ("0", "1"):
        r = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, cwd=root, timeout=600,
                           env=dict(os.environ, VITA_DEBUG=flag))
        assert r.returncode == 0, r.stderr[-2000:]
        outs[flag] = r.stdout.strip().split()
    assert outs["0"][0] == "PLAIN" and outs["1"][0] == "WRAPPED"
    assert outs["0"][1] == outs["1"][1]


@pytest.mark.parametrize("spec", ["te", "local"])
def test_one_node_per_fused_module_is_bit_identical_and_keeps_less(megatron, monkeypatch, spec):
    """NormLinearFn / GatedMLPFn (one autograd node for norm -> linear and for the whole gated MLP; layers.FUSE_AUTOGRAD_NODES) run
    the same kernels in the same order as the node-per-kernel graph: output, input gradient and every matrix gradient bit-identical
    (the norm-weight gradients are fp32 atomic sums: tolerance), with fewer bytes alive between forward and backward."""
    from long_vita_amd import layers
    S = 1024
    ocfg = ollm.LLMConfig(**CFG)
    lp = ollm.init_llm_params(ocfg, seed=77)["layers"][0]
    mcfg = dm.TransformerConfig(hidden_size=CFG["hidden"], num_attention_heads=CFG["heads"], num_query_groups=CFG["kv_groups"],
                                kv_channels=CFG["head_dim"], ffn_hidden_size=CFG["ffn"])
    builder = megatron.get_gpt_layer_with_transformer_engine_spec if spec == "te" else megatron.get_gpt_layer_local_spec
    g = torch.Generator().manual_seed(78)
    x = (torch.randn(S, 1, CFG["hidden"], generator=g) * 0.5).bfloat16().to(DEV)
    go = torch.randn(S, 1, CFG["hidden"], generator=g).bfloat16().to(DEV)
    freqs = glue.rope_emb(S, glue.rope_inv_freq(ocfg.head_dim, ocfg.rope_theta)).to(DEV)
    res = {}
    for fused in (False, True):
        monkeypatch.setattr(layers, "FUSE_AUTOGRAD_NODES", fused)
        layer = dm.build_module(builder(), config=mcfg, layer_number=1)
        _load(layer, lp, spec == "te")
        xh = x.clone().requires_grad_(True)
        held = {}                                  # storage -> bytes of everything autograd saves for the backward (views keep their storage)

        def pack(t):
            st = t.untyped_storage()
            held[st.data_ptr()] = st.nbytes()
            return t
        with torch.autograd.graph.saved_tensors_hooks(pack, lambda t: t):
            out, _ = layer(xh, attention_mask=None, rotary_pos_emb=freqs)
        kept = sum(held.values())                  # (allocator deltas depend on when Python collects the previous layer: not used)
        out.backward(go)
        grads = {n: q.grad.clone() for n, q in layer.named_parameters()}
        res[fused] = (out.detach().clone(), xh.grad.clone(), grads, kept)
        del layer, out, xh
    (o0, dx0, g0, kept0), (o1, dx1, g1, kept1) = res[False], res[True]
    assert torch.equal(o0, o1) and torch.equal(dx0, dx1)
    for n in g0:
        if "norm" in n:
            tol(n, rel_l2(g1[n], g0[n]), 1e-2)
        else:
            assert torch.equal(g0[n], g1[n]), n
    # the gated activation [S, ffn] (and, TE spec, the two normed copies [S, hidden]) are no longer kept
    saved = S * CFG["ffn"] * 2 + (2 * S * CFG["hidden"] * 2 if spec == "te" else 0)
    assert kept0 - kept1 >= 0.9 * saved, (kept0, kept1, saved)


def test_whole_model_under_the_reference_forward_and_loss_func_trains_like_the_oracle(megatron):
    """The composition `pretrain_long_vita.py` runs — forward_step (:841-869) -> GPTVLModel.forward with labels and a logit_mask
    (gpt_vl_model.py:233-416: vision tower -> embedding + visual-token scatter -> RotaryEmbedding -> TransformerBlock -> `unused` ->
    logits-masked output layer -> instruction shift -> vocab-parallel CE) -> loss_func (:778-839) -> autograd — restated in
    tests/dummy_megatron.py with every class looked up under the name the reference imports it from, i.e. built from what the adaptor
    registered: loss and EVERY gradient (embedding table, two full decoder layers, final norm, output layer, projector + its LayerNorm;
    ViT frozen as in stage 3) against the CPU oracle's autograd on the same weights, and against training.TrainStep's explicit sweep."""
    from oracle import train as otrain, vit as ovit
    from long_vita_amd import gpt_vl_model as G, synthetic, training, vision as V
    cfgd = dict(num_layers=2, hidden=1024, heads=8, kv_groups=2, head_dim=128, ffn=2816, vocab=1024)
    ocfg = ollm.LLMConfig(**cfgd)
    p = ollm.init_llm_params(ocfg, seed=8)
    gen = torch.Generator().manual_seed(4)
    p["final_ln"] = (1 + 0.1 * torch.randn(p["final_ln"].shape, generator=gen)).to(p["final_ln"].dtype)
    vcfg = ovit.ViTConfig(num_layers=1, llm_hidden=cfgd["hidden"])
    vp = _randomised_vit_params(ovit, vcfg, seed=9)
    S, n_frames = 768, 2
    tokens, ext = synthetic.make_request(S, n_frames, seed=3, device="cpu")
    tokens = tokens % cfgd["vocab"]
    labels = torch.randint(0, cfgd["vocab"], (1, S), generator=gen)
    loss_mask = torch.zeros(1, S)
    loss_mask[0, S - 100:] = 1
    images = ext["images"]

    # ---- oracle: ViT frozen, projector + decoder differentiable ---------------------------------------------------------------------
    with torch.no_grad():
        x = ovit.vit_embed(images, vp, vcfg)
        for lp in vp["layers"]:
            x = ovit.vit_layer(x, lp, vcfg)
    proj_keys = ("proj_ln_w", "proj_ln_b", "proj_fc1", "proj_fc2")
    pall = dict(p)
    for k in proj_keys:
        pall[k] = vp[k]

    def feature_fn(pp):
        q = dict(vp)
        for k in proj_keys:
            q[k] = pp[k]
        return ovit.vit_project(x, q, vcfg)
    loss_ref, g_ref = otrain.loss_and_grads(tokens, labels, loss_mask, pall, ocfg, feature_fn=feature_fn, indices=ext["indices"])

    # ---- the reference's composition over the registered classes ----------------------------------------------------------------------
    mcfg = dm.TransformerConfig(num_layers=cfgd["num_layers"], hidden_size=cfgd["hidden"], num_attention_heads=cfgd["heads"],
                                num_query_groups=cfgd["kv_groups"], kv_channels=cfgd["head_dim"], ffn_hidden_size=cfgd["ffn"])
    efm, _, model = _vision_model(ovit, vcfg, vp, vit_grad=False, llm_cfg=mcfg, gpt_kwargs=dict(
        transformer_layer_spec=megatron.get_gpt_layer_with_transformer_engine_spec(), vocab_size=cfgd["vocab"], max_sequence_length=S,
        position_embedding_type="rope", rotary_base=ocfg.rope_theta))
    model.embedding.load_state_dict({"word_embeddings.weight": p["embed"].to(DEV)})
    for i, lp in enumerate(p["layers"]):
        _load(model.decoder.layers[i], lp, True)
    model.decoder.final_layernorm.load_state_dict({"weight": p["final_ln"].to(DEV)})
    model.output_layer.load_state_dict({"weight": p["lm_head"].to(DEV)})
    model.unused.data = model.unused.data.to(DEV).bfloat16()
    model.train()
    from long_vita_amd import layers
    assert isinstance(model.decoder.final_layernorm, layers.RMSNorm) and isinstance(model.output_layer, layers.ColumnParallelLinear)
    position_ids = torch.arange(S, dtype=torch.long).unsqueeze(0).to(DEV)
    batch = (tokens.to(DEV), labels.to(DEV), loss_mask.to(DEV), None, position_ids,
             {"images": images.to(DEV).bfloat16(), "indices": ext["indices"].to(DEV)})
    output_tensor, lf = dm.forward_step(batch, model)
    assert output_tensor.dtype == torch.float32 and tuple(output_tensor.shape) == (1, 99)      # 100 answer tokens, shifted by one
    loss_sum, n_tok = lf(output_tensor)
    loss = loss_sum / n_tok                            # schedules.forward_step: output_tensor / num_tokens
    loss.backward()
    assert int(n_tok) == 99
    tol("loss vs oracle", abs(float(loss) - float(loss_ref)) / abs(float(loss_ref)), 2e-2)
    names = {"qkv_w": "self_attention.linear_qkv.weight", "qkv_b": "self_attention.linear_qkv.bias",
             "o_w": "self_attention.linear_proj.weight", "fc1_w": "mlp.linear_fc1.weight", "fc2_w": "mlp.linear_fc2.weight",
             "ln1": "self_attention.linear_qkv.layer_norm_weight", "ln2": "mlp.linear_fc1.layer_norm_weight"}
    got = {"embed": model.embedding.word_embeddings.weight.grad, "final_ln": model.decoder.final_layernorm.weight.grad,
           "lm_head": model.output_layer.weight.grad,
           "proj_ln_w": efm.pre_proj_layernorm.weight.grad, "proj_ln_b": efm.pre_proj_layernorm.bias.grad,
           "proj_fc1": efm.vision_projection.encoder.linear_fc1.weight.grad, "proj_fc2": efm.vision_projection.encoder.linear_fc2.weight.grad}
    errs = {k: rel_l2(v, g_ref[k]) for k, v in got.items()}
    for i, rl in enumerate(g_ref["layers"]):
        params = dict(model.decoder.layers[i].named_parameters())
        for k, n in names.items():
            assert params[n].grad is not None, (i, n)
            errs[f"layers.{i}.{k}"] = rel_l2(params[n].grad, rl[k])
    assert model.unused.grad is not None and float(model.unused.grad.abs().max()) == 0.0     # `hidden_states += 0.0 * self.unused`
    assert all(q.grad is None for n_, q in efm.named_parameters() if n_.startswith("vit."))  # frozen (stage 3)
    _record("whole_model_vs_oracle", errs)
    worst = max(errs, key=errs.get)
    tol(f"worst gradient vs oracle ({worst})", errs[worst], 1.8e-2)

    # ---- and against the explicit sweep on the same weights -----------------------------------------------------------------------------
    vis = V.MegatronVisionModel.from_oracle_layout(V.VisionConfig(num_layers=1, llm_hidden=cfgd["hidden"]), vp, DEV)
    alone = G.GPTVLModel.from_oracle_layout(G.GPTConfig(**cfgd), p, vis, DEV)
    loss2, g2 = training.TrainStep(alone).forward_backward(tokens.to(DEV), labels.to(DEV), loss_mask.to(DEV),
                                                            {"images": images.to(DEV), "indices": ext["indices"].to(DEV)})
    tol("loss vs TrainStep", abs(float(loss) - float(loss2)) / abs(float(loss2)), 5e-3)
    errs2 = {k: rel_l2(got[k], g2[k]) for k in ("embed", "final_ln", "lm_head")}
    errs2.update({k: rel_l2(got[k], g2["projector"][k]) for k in proj_keys})
    for i, gl in enumerate(g2["layers"]):
        params = dict(model.decoder.layers[i].named_parameters())
        for k, n in names.items():
            errs2[f"layers.{i}.{k}"] = rel_l2(params[n].grad, gl[k])
    _record("whole_model_vs_trainstep", errs2)
    worst = max(errs2, key=errs2.get)
    tol(f"worst gradient vs TrainStep ({worst})", errs2[worst], 1.5e-2)

    # ---- the inference branch (labels = None, :357-370): logits [b, n_sel, V] of the masked rows == the stand-alone prefill ---------------
    model.eval()
    lm = loss_mask.bool().to(DEV)
    ext_d = {"images": images.to(DEV).bfloat16(), "indices": ext["indices"].to(DEV)}
    with torch.no_grad():
        lg_mod = model(tokens.to(DEV), position_ids, None, external_inputs=ext_d, logit_mask=lm)
        lg_alone = alone(tokens.to(DEV), position_ids, None, external_inputs=ext_d, logit_mask=lm)
    assert tuple(lg_mod.shape) == tuple(lg_alone.shape) == (1, 100, cfgd["vocab"])
    # two bf16 chains over the same kernels with different fusion points (the driver folds residuals / SwiGLU into GEMM epilogues):
    # measured 7.6e-3 — the distance either keeps from the oracle at this size (smoke: 7.5e-3)
    tol("inference logits, module composition vs stand-alone driver", rel_l2(lg_mod, lg_alone), 1.2e-2)
    assert float((lg_mod.float().argmax(-1) == lg_alone.float().argmax(-1)).float().mean()) >= 0.9


@pytest.mark.parametrize("answers", ["on_both_ranks", "on_rank_0_only"])
def test_whole_model_under_context_parallelism_through_the_reference_composition(megatron, monkeypatch, answers):
    """The same composition on two simulated CP ranks (threads; the collectives of test_train_gpu._run_grid): the reference's get_batch
    (get_batch_on_this_cp_rank + the `external_` prefix stripped, pretrain_long_vita.py:671-696) -> forward_step -> GPTVLModel.forward:
    each rank encodes the frames its chunks hold, scatters its columns (src / tgt indices), rotates with its slice of the angle table,
    runs the decoder with K / V all-gathered (FlashAttnCPFn) and selects ITS answer rows; loss_func all-reduces (sum, tokens) over the CP
    group and scales by CP; the gradients of the two ranks are averaged as Megatron's DDP does.  == the unsharded oracle with the
    reference's per-rank instruction shift.  "on_rank_0_only" = config 5's situation: the answer sits at the end of the row, rank 1's
    logit mask selects nothing — its head, CE and loss_func run on empty tensors and its backward still takes part in every collective."""
    from oracle import train as otrain, vit as ovit
    from long_vita_amd import parallel_state as mpu, synthetic, training_utils
    from test_train_gpu import _run_grid
    cp = 2
    cfgd = dict(num_layers=2, hidden=1024, heads=8, kv_groups=2, head_dim=128, ffn=2816, vocab=1024)
    ocfg = ollm.LLMConfig(**cfgd)
    p = ollm.init_llm_params(ocfg, seed=18)
    vcfg = ovit.ViTConfig(num_layers=1, llm_hidden=cfgd["hidden"])
    vp = _randomised_vit_params(ovit, vcfg, seed=19)
    S, n_frames = 1024, 2
    tokens, ext = synthetic.make_request(S, n_frames, seed=5, device="cpu")
    tokens = tokens % cfgd["vocab"]
    gen = torch.Generator().manual_seed(6)
    labels = torch.randint(0, cfgd["vocab"], (1, S), generator=gen)
    loss_mask = torch.zeros(1, S)
    loss_mask[0, S - 100:] = 1                                    # chunk 3: CP rank 0
    if answers == "on_both_ranks":
        loss_mask[0, 600:640] = 1                                 # chunk 2: CP rank 1
    images = ext["images"]
    with torch.no_grad():
        x = ovit.vit_embed(images, vp, vcfg)
        for lp in vp["layers"]:
            x = ovit.vit_layer(x, lp, vcfg)
    proj_keys = ("proj_ln_w", "proj_ln_b", "proj_fc1", "proj_fc2")
    pall = dict(p)
    for k in proj_keys:
        pall[k] = vp[k]

    def feature_fn(pp):
        q = dict(vp)
        for k in proj_keys:
            q[k] = pp[k]
        return ovit.vit_project(x, q, vcfg)
    loss_ref, g_ref = otrain.loss_and_grads(tokens, labels, loss_mask, pall, ocfg, cp_size=cp, feature_fn=feature_fn, indices=ext["indices"])

    names = {"qkv_w": "self_attention.linear_qkv.weight", "qkv_b": "self_attention.linear_qkv.bias",
             "o_w": "self_attention.linear_proj.weight", "fc1_w": "mlp.linear_fc1.weight", "fc2_w": "mlp.linear_fc2.weight",
             "ln1": "self_attention.linear_qkv.layer_norm_weight", "ln2": "mlp.linear_fc1.layer_norm_weight"}
    monkeypatch.setattr(dm.ARGS, "context_parallel_size", cp)

    def rank_fn(ci, ti):
        with torch.autograd.set_multithreading_enabled(False):
            return rank_body(ci)

    def rank_body(ci):
        mcfg = dm.TransformerConfig(num_layers=cfgd["num_layers"], hidden_size=cfgd["hidden"], num_attention_heads=cfgd["heads"],
                                    num_query_groups=cfgd["kv_groups"], kv_channels=cfgd["head_dim"], ffn_hidden_size=cfgd["ffn"],
                                    context_parallel_size=cp)
        efm, _, model = _vision_model(ovit, vcfg, vp, vit_grad=False, llm_cfg=mcfg, gpt_kwargs=dict(
            transformer_layer_spec=megatron.get_gpt_layer_with_transformer_engine_spec(), vocab_size=cfgd["vocab"], max_sequence_length=S,
            position_embedding_type="rope", rotary_base=ocfg.rope_theta))
        model.embedding.load_state_dict({"word_embeddings.weight": p["embed"].to(DEV)})
        for i, lp in enumerate(p["layers"]):
            _load(model.decoder.layers[i], lp, True)
        model.decoder.final_layernorm.load_state_dict({"weight": p["final_ln"].to(DEV)})
        model.output_layer.load_state_dict({"weight": p["lm_head"].to(DEV)})
        model.unused.data = model.unused.data.to(DEV).bfloat16()
        model.train()
        batch = {"tokens": tokens.to(DEV), "labels": labels.to(DEV), "loss_mask": loss_mask.to(DEV),
                 "position_ids": torch.arange(S, dtype=torch.long, device=DEV).unsqueeze(0),
                 "external_images": images.to(DEV).bfloat16(), "external_indices": ext["indices"].to(DEV)}
        batch = training_utils.get_batch_on_this_cp_rank(batch, seq_length=S)                   # get_batch (:683-692)
        external_inputs = {k[len("external_"):]: batch.pop(k) for k in list(batch) if "external_" in k}
        assert set(external_inputs) == {"images", "src_indices", "tgt_indices"}
        out, lf = dm.forward_step((batch["tokens"], batch["labels"], batch["loss_mask"], None, batch["position_ids"], external_inputs), model)
        loss_x_cp, n_tok = lf(out)
        (loss_x_cp / n_tok).backward()                                                           # schedules.forward_step: / num_tokens
        got = {"embed": model.embedding.word_embeddings.weight.grad, "final_ln": model.decoder.final_layernorm.weight.grad,
               "lm_head": model.output_layer.weight.grad,
               "proj_ln_w": efm.pre_proj_layernorm.weight.grad, "proj_ln_b": efm.pre_proj_layernorm.bias.grad,
               "proj_fc1": efm.vision_projection.encoder.linear_fc1.weight.grad, "proj_fc2": efm.vision_projection.encoder.linear_fc2.weight.grad}
        for i in range(cfgd["num_layers"]):
            params = dict(model.decoder.layers[i].named_parameters())
            for k, n in names.items():
                got[f"layers.{i}.{k}"] = params[n].grad
        assert all(v is not None for v in got.values()), [k for k, v in got.items() if v is None]
        return float(loss_x_cp) / cp / int(n_tok), int(n_tok), int(out.shape[1]), {k: v.detach().float().clone() for k, v in got.items()}

    outs = _run_grid(1, cp, rank_fn, {"mpu": mpu}, monkeypatch)
    n_sel = [int(loss_mask[0, training_utils.zigzag_slice(torch.arange(S), cp, ci, seq_dim=0)].sum()) for ci in range(cp)]
    assert [outs[(ci, 0)][2] for ci in range(cp)] == [max(n - 1, 0) for n in n_sel], (n_sel, [outs[(ci, 0)][2] for ci in range(cp)])
    if answers == "on_rank_0_only":
        assert n_sel[1] == 0
    assert outs[(0, 0)][1] == outs[(1, 0)][1] == sum(max(n - 1, 0) for n in n_sel)              # the all-reduced token count
    tol("loss vs oracle (cp = 2)", abs(outs[(0, 0)][0] - float(loss_ref)) / abs(float(loss_ref)), 2e-2)
    assert outs[(0, 0)][0] == outs[(1, 0)][0]
    errs = {}
    for k in outs[(0, 0)][3]:
        g = sum(outs[(ci, 0)][3][k] for ci in range(cp)) / cp                                    # DDP: the average over the DP x CP group
        ref = g_ref["layers"][int(k.split(".")[1])][k.split(".")[2]] if k.startswith("layers.") else g_ref[k]
        errs[k] = rel_l2(g, ref)
    _record("whole_model_cp2_" + answers, errs)
    worst = max(errs, key=errs.get)
    tol(f"worst gradient vs oracle ({worst})", errs[worst], 2.0e-2)


def test_vit_layer_under_tensor_parallelism_matches_the_unsharded_oracle(megatron, monkeypatch):
    """BASELINE config 5 trains with TP = 2, and the vision tower is built from the same (tensor-parallel) Megatron modules as the decoder
    (the vision args keep `tensor_model_parallel_size`, switch `sequence_parallel` off: M/pretrain_long_vita.py:67): the InternViT layer
    from `get_vit_layer_local_spec_for_intern()` on two simulated TP ranks — qkv / fc1 column-parallel (8 of 16 heads, 2048 of 4096 FFN
    rows per rank, input-gradient all-reduce), proj / fc2 row-parallel (output all-reduce, the bias added once after it) — output and input
    gradient on every rank, sharded gradients reassembled, replicated ones (norms, LayerScale, row-parallel biases) equal on both ranks,
    vs torch autograd over the unsharded oracle layer."""
    from oracle import vit as ovit
    from long_vita_amd import parallel_state as mpu
    from test_train_gpu import _run_grid
    tp = 2
    vls = sys.modules["long_vita_megatron.core.models.vision.vit_layer_specs"]
    vcfg = ovit.ViTConfig(num_layers=1, unfused_bias=True)
    vp = ovit.init_vit_params(vcfg, seed=71)
    gen = torch.Generator().manual_seed(72)
    lp = {k: v.clone() for k, v in vp["layers"][0].items()}
    for k in ("ln1_w", "ln2_w"):
        lp[k] = (1 + 0.1 * torch.randn(lp[k].shape, generator=gen)).bfloat16()
    for k in ("ln1_b", "ln2_b", "proj_b", "fc2_b"):
        lp[k] = (0.1 * torch.randn(lp[k].shape, generator=gen)).bfloat16()
    lp["ls1"] = (0.1 + 0.02 * torch.randn(lp["ls1"].shape, generator=gen)).bfloat16()
    lp["ls2"] = (0.1 + 0.02 * torch.randn(lp["ls2"].shape, generator=gen)).bfloat16()
    n, S, H = 2, vcfg.seq, vcfg.hidden
    x = (torch.randn(n, S, H, generator=gen) * 0.5).bfloat16()
    go = torch.randn(n, S, H, generator=gen).bfloat16()
    xo = x.clone().requires_grad_(True)
    lpo = {k: v.clone().requires_grad_(True) for k, v in lp.items()}
    ref = ovit.vit_layer(xo, lpo, vcfg)
    ref.backward(go)
    names = {"ln1_w": "input_layernorm.weight", "ln1_b": "input_layernorm.bias", "qkv_w": "self_attention.linear_qkv.weight",
             "qkv_b": "self_attention.linear_qkv.bias", "proj_w": "self_attention.linear_proj.weight",
             "proj_b": "self_attention.linear_proj.bias", "ls1": "ls1", "ln2_w": "pre_mlp_layernorm.weight",
             "ln2_b": "pre_mlp_layernorm.bias", "fc1_w": "mlp.linear_fc1.weight", "fc1_b": "mlp.linear_fc1.bias",
             "fc2_w": "mlp.linear_fc2.weight", "fc2_b": "mlp.linear_fc2.bias", "ls2": "ls2"}
    rows = ("qkv_w", "qkv_b", "fc1_w", "fc1_b")          # column-parallel: output features (rows of the weight) split; heads are contiguous
    cols = ("proj_w", "fc2_w")                           # row-parallel: input features split

    def shard(k, t, ti):
        if k in rows:
            return t.chunk(tp, 0)[ti]
        if k in cols:
            return t.chunk(tp, 1)[ti]
        return t

    def rank_fn(ci, ti):
        with torch.autograd.set_multithreading_enabled(False):
            mcfg = dm.TransformerConfig(hidden_size=H, num_attention_heads=vcfg.heads, num_query_groups=vcfg.heads, kv_channels=vcfg.head_dim,
                                        ffn_hidden_size=vcfg.ffn, normalization="LayerNorm", layernorm_epsilon=vcfg.ln_eps,
                                        add_bias_linear=True, add_qkv_bias=True, gated_linear_unit=False,
                                        activation_func=torch.nn.functional.gelu, tensor_model_parallel_size=tp, sequence_parallel=False)
            layer = dm.build_module(vls.get_vit_layer_local_spec_for_intern(), config=mcfg, layer_number=1)
            assert tuple(layer.self_attention.linear_qkv.weight.shape) == (3 * H // tp, H)
            assert tuple(layer.mlp.linear_fc2.weight.shape) == (H, vcfg.ffn // tp)
            layer.load_state_dict({v: shard(k, lp[k], ti).contiguous().to(DEV) for k, v in names.items()})
            xh = x.transpose(0, 1).contiguous().to(DEV).requires_grad_(True)
            out, _ = layer(xh, attention_mask=None)
            out.backward(go.transpose(0, 1).contiguous().to(DEV))
            params = dict(layer.named_parameters())
            return out.detach(), xh.grad.detach(), {k: params[v].grad.detach().clone() for k, v in names.items()}

    outs = _run_grid(tp, 1, rank_fn, {"mpu": mpu}, monkeypatch)
    (o0, dx0, g0), (o1, dx1, g1) = outs[(0, 0)], outs[(0, 1)]
    assert torch.equal(o0, o1) and torch.equal(dx0, dx1)                             # all-reduced: the same on both ranks
    errs = {"out": rel_l2(o0.transpose(0, 1), ref), "dx": rel_l2(dx0.transpose(0, 1), xo.grad)}
    for k in names:
        if k in rows:
            g = torch.cat([g0[k], g1[k]], 0)
        elif k in cols:
            g = torch.cat([g0[k], g1[k]], 1)
        else:
            tol(f"replicated gradient {k}: rank 0 vs rank 1", rel_l2(g0[k], g1[k]), 1e-2)     # fp32 atomic sums: not bit-equal
            g = g0[k]
        errs[k] = rel_l2(g, lpo[k].grad)
    _record("vit_layer_tp2", errs)
    tol("forward", errs["out"], 2.0e-3)
    tol("worst gradient", max(errs.values()), 9e-3)
