"""CPU-side checks (no GPU): the C-ABI library builds, loads and exports every symbol the header
declares; host logic (patch manager, parallel state, sync_output over gloo world_size 2); oracle
self-consistency (restated decoder == transformers Qwen2, zig-zag CP == monolithic)."""
import os
import re
import socket
import subprocess
import sys
import types

import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


# ---------------------------------------------------------------------------------------------
def test_library_builds_and_exports_every_declared_symbol():
    from long_vita_amd import lib
    lib.build()
    declared = set(re.findall(r"\b(vita_[a-z0-9_]+)\s*\(", open(lib.HEADER_PATH).read()))
    declared -= {"vita_attn_params", "vita_attn_bwd_params", "vita_decode_layer_params", "vita_cp_attn_params", "vita_cp_context"}
    assert declared == set(lib.PROTOTYPES), declared ^ set(lib.PROTOTYPES)
    handle = lib.load()                                  # resolves + type-annotates every symbol
    assert handle.vita_abi_version() == lib.ABI_VERSION
    assert handle.vita_error_string(-2).decode().startswith("shape")
    out = subprocess.run(["nm", "-D", "--defined-only", lib.LIB_PATH], capture_output=True, text=True).stdout
    exported = set(re.findall(r" T (vita_[a-z0-9_]+)", out))
    assert declared <= exported


def test_product_path_has_no_cpu_fallback_and_does_not_import_oracle():
    from long_vita_amd import ops
    with pytest.raises(RuntimeError, match="no CPU fallback"):
        ops.rmsnorm(torch.zeros(2, 64, dtype=torch.bfloat16), torch.ones(64, dtype=torch.bfloat16))
    pkg = os.path.join(ROOT, "long_vita_amd")
    for fn in os.listdir(pkg):
        if fn.endswith(".py"):
            src = open(os.path.join(pkg, fn)).read()
            assert not re.search(r"^\s*(from|import)\s+oracle", src, re.M), fn


@pytest.mark.parametrize("struct", ["vita_attn_params", "vita_attn_bwd_params", "vita_decode_layer_params", "vita_cp_attn_params"])
def test_attn_params_struct_matches_header_field_order(struct):
    from long_vita_amd import lib
    hdr = open(lib.HEADER_PATH).read()
    end = hdr.index("} %s;" % struct)
    body = hdr[hdr.rindex("typedef struct {", 0, end): end]
    body = re.sub(r"/\*.*?\*/", "", body, flags=re.S)
    names = []
    for decl in body.split(";"):
        decl = decl.replace("typedef struct {", "").strip()
        if not decl:
            continue
        first, *rest = decl.split(",")
        names.append(re.findall(r"[A-Za-z_][A-Za-z0-9_]*", first)[-1])
        names += [re.findall(r"[A-Za-z_][A-Za-z0-9_]*", r)[-1] for r in rest]
    cls = {"vita_attn_params": lib.AttnParams, "vita_attn_bwd_params": lib.AttnBwdParams,
           "vita_decode_layer_params": lib.DecodeLayerParams, "vita_cp_attn_params": lib.CpAttnParams}[struct]
    assert names == [f[0] for f in cls._fields_]


# ---------------------------------------------------------------------------------------------
def test_patch_manager_semantics():
    from long_vita_amd.patch_utils import MindSpeedPatchesManager as aspm, Patch
    aspm.patches_info = {}
    mod = types.ModuleType("vita_fake_target")
    mod.f = lambda x: x + 1

    class K:
        def m(self, x):
            return x * 2
    mod.K = K
    sys.modules["vita_fake_target"] = mod
    holder = types.ModuleType("vita_fake_holder")
    holder.f = mod.f                                             # `from vita_fake_target import f`
    sys.modules["vita_fake_holder"] = holder

    aspm.register_patch("vita_fake_target.f", lambda x: x + 100)
    with pytest.raises(RuntimeError):
        aspm.register_patch("vita_fake_target.f", lambda x: x)   # second outright patch
    aspm.register_patch("vita_fake_target.f", lambda x: x + 200, force_patch=True)

    def double_wrapper(fn):
        return lambda self, x: fn(self, x) * 10
    aspm.register_patch("vita_fake_target.K.m", double_wrapper)  # name ends with "wrapper" -> decorates
    aspm.register_patch("vita_missing_pkg.sub.g", None, create_dummy=True)
    aspm.apply_patches()
    assert mod.f(1) == 201 and holder.f(1) == 201                # identity-propagated
    assert mod.K().m(3) == 60
    with pytest.raises(RuntimeError, match="no exist"):
        sys.modules["vita_missing_pkg.sub"].g()
    with pytest.raises(ModuleNotFoundError):
        Patch("vita_other_missing.x", lambda: 0, False).apply_patch()
    for k in ("vita_fake_target", "vita_fake_holder", "vita_missing_pkg", "vita_missing_pkg.sub"):
        sys.modules.pop(k, None)
    aspm.patches_info = {}


def test_patch_manager_behaves_like_the_references_patch_utils():
    """The same 30-step script (oracle/make_golden.py:patch_scenario) was run against the reference's M/patch_utils.py
    loaded from its file (fixture patch_manager.pt): outright / forced / wrapper / decorator registration, stacking order,
    class attributes, identity propagation into modules that imported the original (None included), dummy packages,
    missing modules / attributes, and the cumulative re-application quirk — every observable outcome is equal."""
    from conftest import load_golden
    from long_vita_amd.patch_utils import MindSpeedPatchesManager as aspm, Patch
    from oracle.make_golden import patch_scenario
    want = load_golden("patch_manager.pt")["outcomes"]
    got = patch_scenario(aspm, Patch, "vitamirror")
    assert len(want) == 30
    assert got == want, [(a, b) for a, b in zip(got, want) if a != b]


def test_adaptor_registers_reference_targets_with_dummy_megatron():
    """With fabricated megatron modules the adaptor lands its replacements on the reference's dotted
    names (M/megatron_adaptor.py:21-22,93-94,105-106)."""
    import long_vita_amd.megatron_adaptor as ad
    from long_vita_amd.patch_utils import MindSpeedPatchesManager as aspm
    assert not ad.APPLIED                                         # no Megatron in this container
    aspm.patches_info = {}
    try:
        dpa = types.ModuleType("megatron.core.transformer.dot_product_attention")

        class DotProductAttention:
            def forward(self, *a, **k):
                return "orig"
        dpa.DotProductAttention = DotProductAttention
        for name in ["megatron", "megatron.core", "megatron.core.transformer"]:
            sys.modules[name] = types.ModuleType(name)
        sys.modules["megatron.core.transformer.dot_product_attention"] = dpa
        sys.modules["megatron.core.transformer"].dot_product_attention = dpa
        sys.modules["megatron.core"].transformer = sys.modules["megatron.core.transformer"]
        sys.modules["megatron"].core = sys.modules["megatron.core"]
        assert ad.exe_adaptation(create_dummy=True)
        from long_vita_amd.layers import ColumnParallelLinear
        assert sys.modules["megatron.core.tensor_parallel.layers"].ColumnParallelLinear is ColumnParallelLinear
        assert DotProductAttention.forward.__name__ == "forward" and DotProductAttention.forward is not None
        assert DotProductAttention.forward.__wrapped__.__qualname__.endswith("DotProductAttention.forward")
        # the decode loop lands on the reference's target (:143) and takes its switches from Megatron's globals (:71-76)
        gen_mod = sys.modules["megatron.inference.text_generation.generation"]
        assert gen_mod.generate_tokens_probs_and_return_on_first_stage is ad.generate_tokens_probs_and_return_on_first_stage
        V = 13
        nxt = lambda tok, pos: (tok * 7 + pos * 3 + 1) % V
        full = [4, 2, 9]
        while len(full) < 12:
            full.append(int(nxt(torch.tensor(full[-1]), torch.tensor(len(full) - 1))))
        stop = next(i for i in range(5, 12) if full[i] not in full[3:i])          # first generated token that is new
        training = types.ModuleType("megatron.training")
        training.get_args = lambda: types.SimpleNamespace(use_kv_cache=False, logit_mask=True, eos_id=full[stop])
        training.get_tokenizer = lambda: types.SimpleNamespace(eod=-1)
        sys.modules["megatron.training"] = training

        def model(tokens, position_ids, attention_mask, inference_params=None):
            sel = inference_params.logit_mask[0].nonzero().flatten()
            return torch.nn.functional.one_hot(nxt(tokens[0, sel], position_ids[0, sel]), V).float()[None]
        tokens = torch.zeros(1, 12, dtype=torch.long)
        tokens[0, :3] = torch.tensor(full[:3])
        steps = list(gen_mod.generate_tokens_probs_and_return_on_first_stage(model, tokens, torch.tensor([3])))
        assert steps[-1][0][0].tolist() == full[:stop + 1] and stop < 11          # stops at args.eos_id
        # ForwardStep(..., external_inputs=...) (:145): the wrapper pops the keyword and hangs it on the InferenceParams
        fs_mod = sys.modules["megatron.inference.text_generation.forward_step"]

        class UpstreamForwardStep:
            def __init__(self, model, max_batch_size, max_sequence_length):
                self.inference_params = types.SimpleNamespace(external_inputs="unset")
        UpstreamForwardStep.__init__ = ad.inference_forward_step_init_wrapper(UpstreamForwardStep.__init__)
        assert UpstreamForwardStep(None, 1, 8, external_inputs={"images": 1}).inference_params.external_inputs == {"images": 1}
        assert UpstreamForwardStep(None, 1, 8).inference_params.external_inputs is None
        assert fs_mod.ForwardStep.__init__ is not None
    finally:
        for k in [k for k in sys.modules if k == "megatron" or k.startswith("megatron.")]:
            sys.modules.pop(k)
        aspm.patches_info = {}


def test_megatron_constructs_the_hip_modules_through_the_patched_layer_specs():
    """under a (stand-in) Megatron the adaptor's spec builders land on the reference's dotted names
    (M/megatron_adaptor.py:81-88) and `build_module(spec, config=, layer_number=)` — Megatron's own construction calls, with
    Megatron's constructor arguments — yields a decoder layer whose leaves are this package's nn.Modules, with Parameters under
    Megatron's checkpoint names.  CPU construction (`use_cpu_initialization`); the forward needs the GPU (tests/test_boundary_gpu.py)."""
    import dummy_megatron as dm
    import long_vita_amd.megatron_adaptor as ad
    from long_vita_amd import layers
    from long_vita_amd.dot_product_attention import HipDotProductAttention
    from long_vita_amd.language_model_embedding import LanguageModelEmbedding
    from long_vita_amd.patch_utils import MindSpeedPatchesManager as aspm
    aspm.patches_info = {}
    names = dm.install()
    try:
        assert ad.exe_adaptation(create_dummy=True)
        specs = sys.modules["megatron.core.models.gpt.gpt_layer_specs"]
        cfg = dm.TransformerConfig(use_cpu_initialization=True)
        te = dm.build_module(specs.get_gpt_layer_with_transformer_engine_spec(), config=cfg, layer_number=1)
        assert isinstance(te.self_attention.linear_qkv, layers.LayerNormColumnParallelLinear)
        assert isinstance(te.self_attention.core_attention, HipDotProductAttention)
        assert isinstance(te.self_attention.linear_proj, layers.RowParallelLinear)
        assert isinstance(te.mlp.linear_fc1, layers.LayerNormColumnParallelLinear) and isinstance(te.mlp.linear_fc2, layers.RowParallelLinear)
        sd = te.state_dict()
        h, q, kv, f = cfg.hidden_size, cfg.kv_channels * cfg.num_attention_heads, cfg.kv_channels * cfg.num_query_groups, cfg.ffn_hidden_size
        want = {"self_attention.linear_qkv.layer_norm_weight": (h,), "self_attention.linear_qkv.weight": (q + 2 * kv, h),
                "self_attention.linear_qkv.bias": (q + 2 * kv,), "self_attention.linear_proj.weight": (h, q),
                "mlp.linear_fc1.layer_norm_weight": (h,), "mlp.linear_fc1.weight": (2 * f, h), "mlp.linear_fc2.weight": (h, f)}
        got = {k: tuple(v.shape) for k, v in sd.items() if v is not None and not k.endswith("_extra_state")}
        assert got == want, got                                        # the names R/tools/hf2mcore_long_vita.py:590-617 writes
        assert all(isinstance(p_, torch.nn.Parameter) and p_.dtype == torch.bfloat16 for p_ in te.parameters())
        assert float(te.self_attention.linear_qkv.bias.abs().max()) == 0.0 and float(te.mlp.linear_fc1.weight.float().std()) > 0.01
        te.load_state_dict({k: torch.zeros(v) for k, v in want.items()})   # a checkpoint without TE's _extra_state entries loads
        local = dm.build_module(specs.get_gpt_layer_local_spec(), config=cfg, layer_number=1)
        assert isinstance(local.input_layernorm, layers.RMSNorm) and isinstance(local.self_attention.linear_qkv, layers.ColumnParallelLinear)
        assert {"input_layernorm.weight", "pre_mlp_layernorm.weight", "self_attention.linear_qkv.weight", "mlp.linear_fc1.weight"} <= set(local.state_dict())
        # the two classes the reference replaces outright (:93-94,105-106) keep Megatron's constructor signatures
        emb_cls = sys.modules["megatron.core.models.common.embeddings.language_model_embedding"].LanguageModelEmbedding
        assert emb_cls is LanguageModelEmbedding
        emb = emb_cls(config=cfg, vocab_size=256, max_sequence_length=64, position_embedding_type="rope")
        assert list(emb.state_dict()) == ["word_embeddings.weight"] and tuple(emb.word_embeddings.weight.shape) == (256, h)
        cpl_cls = sys.modules["megatron.core.tensor_parallel.layers"].ColumnParallelLinear
        out_layer = cpl_cls(h, 256, config=cfg, init_method=cfg.init_method, bias=False, skip_bias_add=False, gather_output=False,
                            skip_weight_param_allocation=True)       # the shared-embedding output layer of GPTModel
        assert out_layer.weight is None
        with pytest.raises(RuntimeError, match="skip_weight_param_allocation"):
            out_layer(torch.zeros(4, 1, h, dtype=torch.bfloat16))
        with pytest.raises(RuntimeError, match="supplied weight's shape"):
            out_layer(torch.zeros(4, 1, h, dtype=torch.bfloat16), weight=torch.zeros(8, h, dtype=torch.bfloat16))
        with pytest.raises(RuntimeError, match="no CPU fallback"):    # the product path never computes on the host
            te.self_attention.linear_proj(torch.zeros(4, 1, q, dtype=torch.bfloat16))
    finally:
        dm.uninstall(names)
        aspm.patches_info = {}


def test_adaptor_target_names_are_the_references_call_sites():
    """Every dotted name the adaptor registers is one the reference registers (M/megatron_adaptor.py call sites read with
    ast, fixture adaptor_targets.pt) with the same kind of replacement (wrapper vs outright), except the documented extra;
    the reference targets left to Megatron are spelled out."""
    from conftest import load_golden
    import long_vita_amd.megatron_adaptor as ad
    g = load_golden("adaptor_targets.pt")
    live = {t[0]: t[1] for t in g["live_targets"]}
    assert len(live) == 16 and "mcore_parallel_state_adaptation" not in g["called"]          # defined, never called (:158)
    mine = dict(ad._targets())
    for name, obj in mine.items():
        if name in ad.EXTRA_TARGETS:
            continue
        assert name in live, name
        ref_is_wrapper = live[name].endswith(("wrapper", "decorator"))
        assert obj.__name__.endswith(("wrapper", "decorator")) == ref_is_wrapper, name
        assert obj.__name__ == live[name], (name, obj.__name__, live[name])               # same replacement names too
    left = sorted(set(live) - set(mine))
    assert left == sorted([
        "megatron.core.transformer.transformer_config.TransformerConfig",
        "megatron.training.checkpointing.ensure_directory_exists",
        "megatron.inference.text_generation.tokenization.tokenize_prompts",
        "megatron.inference.text_generation.tokenization._tokenize_prompts_and_batch",
        "megatron.inference.text_generation.generation.beam_search_and_return_on_first_stage",
        "megatron.inference.text_generation.forward_step._no_pipelining_forward_step",
        "megatron.inference.text_generation.forward_step._with_pipelining_forward_step",
        "megatron.training.arguments.parse_args", "megatron.training.global_vars.build_tokenizer"])
    assert [a[0] for a in g["assignments"]] == ["megatron.legacy.data.data_samplers.build_pretraining_data_loader",
                                                "megatron.training.training.build_pretraining_data_loader",
                                                "megatron.core.optimizer._get_param_groups"]


# ---------------------------------------------------------------------------------------------
_GLOO_WORKER = r"""
import os, sys, torch, torch.distributed as dist
sys.path.insert(0, os.environ["VITA_ROOT"])
from long_vita_amd import generation, parallel_state as mpu
from oracle import glue
dist.init_process_group("gloo", rank=int(os.environ["RANK"]), world_size=int(os.environ["WORLD_SIZE"]))
mpu.initialize_model_parallel()
cp, r = mpu.get_context_parallel_world_size(), mpu.get_context_parallel_rank()
assert (cp, r) == (dist.get_world_size(), dist.get_rank())
V = 11
# rank r "computed" logits for its two zig-zag halves: value = chunk id
ids = mpu.zigzag_chunk_ids(cp, r)
local = torch.stack([torch.full((V,), float(i)) for i in ids])[None]          # [1, 2, V]
out = generation.sync_output(local)
assert out.shape == (1, 2 * cp, V)
assert torch.equal(out[0, :, 0], torch.arange(2 * cp, dtype=torch.float32)), out[0, :, 0]
assert glue.sync_output_order(cp) == sorted(range(2 * cp), key=lambda j: [i for q in range(cp) for i in mpu.zigzag_chunk_ids(cp, q)][j])
# logit-mask rule vs oracle restatement on every context length
toks = torch.zeros(1, 16, dtype=torch.long)
for ctx in range(1, 16 * cp):
    for compat in (True, False):
        m, blk = generation.build_logit_mask(toks, ctx, compat)
        pos, b2 = glue.cp_logit_mask_positions(ctx, 16, cp, compat)
        assert m[0].nonzero().flatten().tolist() == pos and blk == b2, (ctx, compat)
dist.barrier()
dist.destroy_process_group()
print("OK", r)
"""


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def test_sync_output_and_parallel_state_gloo_world2(tmp_path):
    script = tmp_path / "worker.py"
    script.write_text(_GLOO_WORKER)
    port = _free_port()
    procs = []
    for r in range(2):
        env = dict(os.environ, RANK=str(r), WORLD_SIZE="2", MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port),
                   VITA_ROOT=ROOT)
        procs.append(subprocess.Popen([sys.executable, str(script)], env=env, stdout=subprocess.PIPE,
                                      stderr=subprocess.STDOUT, text=True))
    for p in procs:
        out, _ = p.communicate(timeout=300)
        assert p.returncode == 0 and "OK" in out, out


# ---------------------------------------------------------------------------------------------
def test_oracle_decoder_matches_transformers_qwen2():
    """Restated Megatron-layout decoder (fp32) == transformers Qwen2ForCausalLM on converted weights
    (SURVEY.md §8c cross-checks iv, v: RoPE theta=1e6 convention, RMSNorm, QKV group layout)."""
    from transformers import Qwen2Config, Qwen2ForCausalLM

    from oracle import llm as ollm
    cfg = ollm.LLMConfig(num_layers=2, hidden=256, heads=8, kv_groups=2, head_dim=32, ffn=512, vocab=300)
    p = ollm.init_llm_params(cfg, seed=3, dtype=torch.float32, std=0.05)
    hf = Qwen2ForCausalLM(Qwen2Config(vocab_size=300, hidden_size=256, intermediate_size=512, num_hidden_layers=2,
                                      num_attention_heads=8, num_key_value_heads=2, rms_norm_eps=1e-6,
                                      rope_theta=1e6, max_position_embeddings=4096, tie_word_embeddings=False,
                                      attn_implementation="eager")).eval()
    missing, unexpected = hf.load_state_dict(ollm.to_hf_state_dict(p, cfg), strict=False)
    assert not unexpected and all("rotary" in m or "inv_freq" in m for m in missing), (missing, unexpected)
    tokens = torch.randint(0, 300, (1, 96), generator=torch.Generator().manual_seed(1))
    with torch.no_grad():
        ref = hf(tokens).logits
    out = ollm.prefill_logits(tokens, p, cfg, range(96))
    torch.testing.assert_close(out, ref, rtol=2e-4, atol=2e-4)


def test_oracle_cp_prefill_equals_cp1():
    from oracle import glue, llm as ollm
    cfg = ollm.LLMConfig(num_layers=2, hidden=128, heads=4, kv_groups=2, head_dim=32, ffn=256, vocab=200)
    p = ollm.init_llm_params(cfg, seed=4, dtype=torch.float32, std=0.05)
    S, cp = 64, 4
    tokens = torch.randint(0, 200, (1, S), generator=torch.Generator().manual_seed(2))
    full = ollm.prefill_logits(tokens, p, cfg, range(S))[0]                     # [S, V]
    outs = ollm.prefill_logits_cp(tokens, p, cfg, cp, [range(S // cp)] * cp)
    for r in range(cp):
        torch.testing.assert_close(outs[r][0], glue.zigzag_slice(full[None], cp, r)[0], rtol=1e-4, atol=1e-4)


# ---------------------------------------------------------------------------------------------
@pytest.mark.parametrize("cp,chunk,P", [(2, 4, 16), (2, 4, 9), (2, 4, 3), (4, 2, 11), (4, 2, 16), (1, 8, 5)])
def test_kv_cache_compaction_keeps_exactly_the_real_prompt_rows(cp, chunk, P):
    """After a padded zig-zag prefill the shard keeps the rows whose global position is < P, in position order
    (gpt_vl_model._compact_cache), and the shards of all ranks partition [0, P)."""
    from long_vita_amd import parallel_state as mpu
    from long_vita_amd.gpt_vl_model import GPTVLModel
    from long_vita_amd.inference_params import InferenceParams
    from oracle import glue
    S = 2 * cp * chunk if cp > 1 else chunk
    seen = []
    try:
        for r in range(cp):
            mpu.set_context_parallel_state(cp, r, None)
            pos = torch.arange(S)[None]
            local = glue.zigzag_slice(pos, cp, r)[0] if cp > 1 else pos[0]
            ip = InferenceParams(1, S + 4)
            ip.prefill_valid_tokens = P
            kv = torch.full((2, local.numel() + 3, 1, 1), -1.0)
            kv[:, : local.numel(), 0, 0] = local.float()
            ip.key_value_memory_dict = {1: kv}
            GPTVLModel._compact_cache(object.__new__(GPTVLModel), ip, local.numel())
            want = [int(x) for x in local if int(x) < P]
            assert ip.local_len == len(want) and ip.consumed_tokens == P
            assert kv[0, : ip.local_len, 0, 0].tolist() == want and kv[1, : ip.local_len, 0, 0].tolist() == want
            seen += want
    finally:
        mpu.set_context_parallel_state(1, 0, None)
    assert sorted(seen) == list(range(P))


def test_decode_loop_host_rules():
    from long_vita_amd import generation as gen
    assert gen._cp_prefill_length(1500, 2) == 2048 and gen._cp_prefill_length(1024, 2) == 1024
    assert gen._cp_prefill_length(131072, 8) == 131072 and gen._cp_prefill_length(131073, 8) == 131072 + 4096
    logits = torch.tensor([[0.1, 3.0, -1.0, 2.9]])
    assert gen._sample_strategy(logits)[1].tolist() == [1]
    g = torch.manual_seed(0)                                                  # noqa: F841
    picks = {int(gen._sample_strategy(logits, do_sample=True, top_k=2)[1]) for _ in range(50)}
    assert picks <= {1, 3} and len(picks) == 2
    picks = {int(gen._sample_strategy(logits, do_sample=True, top_p=0.3)[1]) for _ in range(20)}
    assert picks == {1}                                                       # nucleus keeps the top token only


def test_packed_segments_from_position_ids():
    """Packed samples are recognised the way the reference's GPU path does it (non-monotonic position ids, cu_seqlens at
    their zeros) and turned into per-row segment bounds."""
    from long_vita_amd import ops, training_utils as tu
    cuts, S = [0, 5, 6, 20], 32
    pos = torch.cat([torch.arange(b - a) for a, b in zip(cuts, cuts[1:] + [S])])
    try:
        tu.set_position_ids(pos[:, None])
        start, end = tu.get_packed_segments()
        want_start = [max(c for c in cuts if c <= i) for i in range(S)]
        want_end = [min(c for c in cuts[1:] + [S] if c > i) for i in range(S)]
        assert start.tolist() == want_start and end.tolist() == want_end and start.dtype == torch.int32
        tu.set_position_ids(torch.arange(S)[:, None])
        assert tu.get_packed_segments() is None                       # monotonic ids: one sample
        tu.set_position_ids(None)
        assert tu.get_packed_segments() is None
    finally:
        tu.set_position_ids(None)
    s2, e2 = ops.segments_from_cu_seqlens(torch.tensor([0, 10, 25]), 32)   # tail rows form their own segment
    assert s2.tolist() == [0] * 10 + [10] * 15 + [25] * 7 and e2.tolist() == [10] * 10 + [25] * 15 + [32] * 7


def test_packed_segments_match_transformers_varlen_preparation():
    """The third-party step behind the reference's packed path: M/core/transformer/dot_product_attention.py:374-390 calls
    transformers' _flash_attention_forward(position_ids=...), which derives the flash_attn_varlen cu_seqlens with
    prepare_fa_kwargs_from_position_ids (transformers >= 4.48.3 pinned by the reference, 5.x installed).  The segment
    bounds the HIP kernels get are the same partition, for random packings incl. length-1 samples."""
    from transformers.modeling_flash_attention_utils import prepare_fa_kwargs_from_position_ids

    from long_vita_amd import training_utils as tu
    g = torch.Generator().manual_seed(11)
    for _ in range(20):
        S = int(torch.randint(8, 200, (1,), generator=g))
        n_cut = int(torch.randint(1, 6, (1,), generator=g))
        cuts = sorted(set([0] + torch.randint(1, S, (n_cut,), generator=g).tolist()))
        pos = torch.cat([torch.arange(b - a) for a, b in zip(cuts, cuts[1:] + [S])])
        (cu, _), (max_len, _) = prepare_fa_kwargs_from_position_ids(pos[None])
        assert cu.tolist() == cuts + [S]
        try:
            tu.set_position_ids(pos[:, None])
            seg = tu.get_packed_segments()
        finally:
            tu.set_position_ids(None)
        if len(cuts) == 1:
            assert seg is None                                        # monotonic: transformers stays on the dense path too
            continue
        start, end = seg
        for a, b in zip(cu.tolist()[:-1], cu.tolist()[1:]):
            assert start[a:b].tolist() == [a] * (b - a) and end[a:b].tolist() == [b] * (b - a)
        assert int(max_len) == int((end - start).max())


def test_packed_segments_reproduce_the_references_reset_attention_mask():
    """Stage-2 packing end to end on the host side: the reference's get_ltor_masks_and_position_ids (--reset-position-ids
    --reset-attention-mask, M/training/utils.py:192-250) turned token rows with EOD tokens into position ids with resets and
    a block-diagonal causal mask, and compute_actual_seq_len (:53-57) into sample ends (fixture packed_positions.pt).  The
    segment bounds the HIP kernels get from those position ids describe exactly that mask and those ends — EOD as its own
    length-1 sample, consecutive EODs, EOD in the first and in the last position included."""
    from conftest import load_golden
    from long_vita_amd import training_utils as tu
    from oracle.attention import core_attention
    rows = load_golden("packed_positions.pt")["rows"]
    assert len(rows) == 4
    for r in rows:
        pos, ref_mask = r["position_ids"], r["attention_mask"][0, 0]            # True = masked
        S = pos.shape[1]
        try:
            tu.set_position_ids(pos.transpose(0, 1).contiguous())
            seg = tu.get_packed_segments()
        finally:
            tu.set_position_ids(None)
        q, k = torch.arange(S)[:, None], torch.arange(S)[None, :]
        if seg is None:
            assert r["actual_seq_len"] == [S] and torch.equal(ref_mask, k > q)
            continue
        start, end = seg[0].long(), seg[1].long()
        allowed = (k <= q) & (k >= start[:, None])
        assert torch.equal(~ref_mask, allowed)
        assert sorted(set(end.tolist())) == r["actual_seq_len"]
        # and the oracle's varlen rule (what the GPU tests compare the kernels with) is the same mask
        cu = torch.tensor([0] + r["actual_seq_len"], dtype=torch.int32)
        x = torch.randn(S, 1, 2, S, generator=torch.Generator().manual_seed(S))
        v = torch.eye(S)[:, None, None, :].expand(S, 1, 2, S).contiguous()        # values = one-hot keys: output = probabilities
        probs = core_attention(x, x, v, causal=True, cu_seqlens=cu).view(S, 2, S)
        assert torch.equal(probs[:, 0] > 0, allowed)


_TPCP_WORKER = r"""
import os, sys, torch, torch.distributed as dist
sys.path.insert(0, os.environ["VITA_ROOT"])
from long_vita_amd import parallel_state as mpu
dist.init_process_group("gloo", rank=int(os.environ["RANK"]), world_size=int(os.environ["WORLD_SIZE"]))
mpu.initialize_model_parallel(tensor_model_parallel_size=2)          # world 4 -> TP 2 x CP 2, TP ranks adjacent
me = dist.get_rank()
assert (mpu.get_tensor_model_parallel_world_size(), mpu.get_context_parallel_world_size()) == (2, 2)
assert mpu.get_tensor_model_parallel_rank() == me % 2 and mpu.get_context_parallel_rank() == me // 2
t = torch.tensor([float(me)])
dist.all_reduce(t, group=mpu.get_tensor_model_parallel_group())
assert t.item() == {0: 1.0, 1: 1.0, 2: 5.0, 3: 5.0}[me]                # {0,1} and {2,3}
c = torch.tensor([float(me)])
dist.all_reduce(c, group=mpu.get_context_parallel_group())
assert c.item() == {0: 2.0, 2: 2.0, 1: 4.0, 3: 4.0}[me]                # {0,2} and {1,3}
dist.barrier()
dist.destroy_process_group()
print("OK", me)
"""


def test_tensor_x_context_parallel_groups_gloo_world4(tmp_path):
    script = tmp_path / "worker.py"
    script.write_text(_TPCP_WORKER)
    port = _free_port()
    procs = []
    for r in range(4):
        env = dict(os.environ, RANK=str(r), WORLD_SIZE="4", MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), VITA_ROOT=ROOT)
        procs.append(subprocess.Popen([sys.executable, str(script)], env=env, stdout=subprocess.PIPE,
                                      stderr=subprocess.STDOUT, text=True))
    for p in procs:
        out, _ = p.communicate(timeout=300)
        assert p.returncode == 0 and "OK" in out, out


_TP_BATCH_WORKER = r"""
import os, sys, torch, torch.distributed as dist
sys.path.insert(0, os.environ["VITA_ROOT"])
from long_vita_amd import parallel_state as mpu, training_utils as tu
from oracle.make_golden import TP_BATCH_CASES, tp_batch_data
g = torch.load(os.path.join(os.environ["VITA_ROOT"], "tests", "golden", "tp_batch.pt"), weights_only=True)
rank = int(os.environ["RANK"])
dist.init_process_group("gloo", rank=rank, world_size=2)
mpu.initialize_model_parallel(tensor_model_parallel_size=2, context_parallel_size=1)
assert mpu.get_tensor_model_parallel_src_rank() == 0
for case, ref in zip(TP_BATCH_CASES, g["cases"]):
    tu.set_actual_seq_len(None)
    it = iter(tp_batch_data(case)) if rank == 0 else None
    b = tu.get_batch_on_this_tp_rank(it, micro_batch_size=2, seq_length=64, image_size=28, reset_attention_mask=case["reset"])
    want = ref["ranks"][rank]
    for k, v in want.items():
        if k == "actual_seq_len":
            assert tu.get_actual_seq_len() == v, (case["name"], tu.get_actual_seq_len(), v)
        elif v is None:
            assert b.get(k) is None, (case["name"], k)
        else:
            assert b[k].dtype == v.dtype and torch.equal(b[k], v), (case["name"], k)
    assert set(b) == set(k for k in want if k != "actual_seq_len"), (case["name"], sorted(b))
dist.barrier()
dist.destroy_process_group()
print("OK", rank)
"""


def test_get_batch_on_this_tp_rank_matches_the_references_own_function(tmp_path):
    """VERDICT r1 missing #3: the tensor-parallel batch broadcast (M/training/utils.py:410-626) on two gloo ranks against a
    fixture produced by the reference's own function on two gloo ranks (oracle/make_golden.py:golden_tp_batch): same keys,
    dtypes (frames bf16, indices int64) and values on both ranks; a batch without images gets the all-ones placeholder frame;
    `actual_seq_len` travels through the dynamic broadcast."""
    script = tmp_path / "worker.py"
    script.write_text(_TP_BATCH_WORKER)
    port = _free_port()
    procs = []
    for r in range(2):
        env = dict(os.environ, RANK=str(r), WORLD_SIZE="2", MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), VITA_ROOT=ROOT)
        procs.append(subprocess.Popen([sys.executable, str(script)], env=env, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True))
    for p in procs:
        out, _ = p.communicate(timeout=300)
        assert p.returncode == 0 and "OK" in out, out


def test_sample_strategy_matches_the_references_own_functions():
    """_sample_strategy against the reference's _sample_strategy + top_k_logits executed from source (fixture sampling.pt):
    the filtered distribution bit for bit (temperature, top-k incl. k = 1 and ties at the k-th value, top-p incl. p = 1 and
    the kept first-above-threshold token), the token drawn under the same torch seed, the greedy branch, and — unlike an
    fp32 input in the reference — an untouched caller tensor."""
    from conftest import load_golden
    from long_vita_amd.generation import _sample_strategy
    from oracle.make_golden import SAMPLING_CASES, sampling_case_logits
    g = load_golden("sampling.pt")["cases"]
    assert len(g) == len(SAMPLING_CASES)
    for i, c in enumerate(g):
        logits = sampling_case_logits(i, c).bfloat16()
        keep = logits.clone()
        torch.manual_seed(5 + i)
        probs, tok = _sample_strategy(logits, True, top_k=c["top_k"], top_p=c["top_p"], temperature=c["temperature"])
        assert torch.equal(probs, c["probs"]), i
        assert torch.equal(tok, c["token"]), i
        same, greedy = _sample_strategy(logits, False)
        assert same is logits and torch.equal(greedy, c["greedy"]) and torch.equal(logits, keep)
        f32 = logits.float()
        _sample_strategy(f32, True, top_k=c["top_k"], top_p=c["top_p"], temperature=2.0)
        assert torch.equal(f32, keep.float())


_DECODE_WORKER = r"""
import os, sys, types, torch, torch.distributed as dist
sys.path.insert(0, os.environ["VITA_ROOT"])
from long_vita_amd import generation, parallel_state as mpu
case = torch.load(os.environ["VITA_CASE"])
cp, seq, V = case["cp"], case["seq"], case["vocab"]
rank = int(os.environ["RANK"])
if cp > 1:
    dist.init_process_group("gloo", rank=rank, world_size=cp)
mpu.initialize_model_parallel()
nxt = lambda tok, pos: (tok * 7 + pos * 3 + 1) % V               # oracle/make_golden.py:fake_next_token
masks = []

def model(tokens, position_ids, attention_mask, inference_params=None):
    sel = inference_params.logit_mask[0].nonzero().flatten()
    masks.append(sel.tolist())
    return torch.nn.functional.one_hot(nxt(tokens[0, sel], position_ids[0, sel]), V).float()[None]

for compat in (True, False):
    masks.clear()
    tokens = case["prompt"].clone()
    for _ in generation.generate_tokens_probs_and_return_on_first_stage(model, tokens, case["lengths"].clone(), use_kv_cache=False,
                                                                       logit_mask=True, reference_compat=compat):
        pass
    if compat:      # bit for bit what the reference's own loop did on this rank: every mask, every generated token
        assert masks == case["masks"][rank], (masks[:4], case["masks"][rank][:4])
        assert torch.equal(tokens, case["tokens"])
    else:           # the corrected rule follows the true continuation through every chunk boundary
        n = int(case["lengths"][0])
        for ctx in range(n, seq):
            assert int(tokens[0, ctx]) == int(nxt(tokens[0, ctx - 1], torch.tensor(ctx - 1))), ctx
if cp > 1:
    dist.barrier()
    dist.destroy_process_group()
print("OK", rank)
"""


@pytest.mark.parametrize("idx", [0, 1, 2])
def test_decode_loop_matches_the_references_own_loop(tmp_path, idx):
    """generate_tokens_probs_and_return_on_first_stage (no-cache CP path, M/inference/text_generation/generation.py:33-280)
    against a fixture made by running the REFERENCE's loop — with its get_batch_on_this_cp_rank and sync_output — on CP gloo
    ranks under a stub Megatron (oracle/make_golden.py:golden_decode_loop): the same logit-mask positions on every rank at
    every step and the same generated tokens, including the reference's off-by-one block at ctx % (S/2CP) == 0
    (its fixture leaves the true continuation exactly there); reference_compat=False stays on it."""
    from conftest import load_golden
    case = load_golden("decode_loop.pt")["cases"][idx]
    path = tmp_path / "case.pt"
    torch.save(case, path)
    script = tmp_path / "worker.py"
    script.write_text(_DECODE_WORKER)
    port = _free_port()
    procs = []
    for r in range(case["cp"]):
        env = dict(os.environ, RANK=str(r), WORLD_SIZE=str(case["cp"]), MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port),
                   VITA_ROOT=ROOT, VITA_CASE=str(path))
        procs.append(subprocess.Popen([sys.executable, str(script)], env=env, stdout=subprocess.PIPE,
                                      stderr=subprocess.STDOUT, text=True))
    for p in procs:
        out, _ = p.communicate(timeout=300)
        assert p.returncode == 0 and "OK" in out, out


def test_tensor_parallel_shards_roundtrip():
    from long_vita_amd import tensor_parallel as tpar
    from oracle import llm as ollm
    cfg = ollm.LLMConfig(num_layers=2, hidden=256, heads=8, kv_groups=4, head_dim=32, ffn=512, vocab=320)
    p = ollm.init_llm_params(cfg, seed=2)
    shards = [tpar.shard_llm_params(p, cfg, 2, r) for r in range(2)]
    assert shards[0][1].heads == 4 and shards[0][1].kv_groups == 2 and shards[0][1].ffn == 256
    full = tpar.unshard_llm_grads([s[0] for s in shards], cfg, 2)
    for k in ("embed", "final_ln", "lm_head"):
        assert torch.equal(full[k], p[k])
    for a, b in zip(full["layers"], p["layers"]):
        assert all(torch.equal(a[k], b[k]) for k in b)
    # the shards compute the same function: column-parallel qkv rows are whole kv groups, fc1 keeps gate / up pairing
    x = torch.randn(5, 256)
    y_full = torch.nn.functional.silu(x @ p["layers"][0]["fc1_w"].float()[:512].T) * (x @ p["layers"][0]["fc1_w"].float()[512:].T)
    y_sh = torch.cat([torch.nn.functional.silu(x @ s[0]["layers"][0]["fc1_w"].float()[:256].T)
                      * (x @ s[0]["layers"][0]["fc1_w"].float()[256:].T) for s in shards], dim=1)
    torch.testing.assert_close(y_sh, y_full)


def test_oracle_siglip_matches_transformers():
    """The SigLIP variant of the restated ViT layer (tanh GELU, no LayerScale / class token, 16 x 72 heads, FFN 4304) ==
    transformers' SiglipVisionModel encoder in fp32 (Megatron per-head QKV layout -> q_proj / k_proj / v_proj)."""
    from transformers import SiglipVisionConfig, SiglipVisionModel

    from oracle import vit as ovit
    cfg = ovit.ViTConfig.siglip_400m(num_layers=2, image=56)                  # 4 x 4 patches keep it small
    p = ovit.init_vit_params(cfg, seed=6, dtype=torch.float32)
    hf = SiglipVisionModel(SiglipVisionConfig(hidden_size=1152, intermediate_size=4304, num_hidden_layers=2,
                                              num_attention_heads=16, image_size=56, patch_size=14,
                                              hidden_act="gelu_pytorch_tanh", layer_norm_eps=1e-6,
                                              attn_implementation="eager")).eval()
    sd = hf.state_dict()
    pre = "vision_model." if any(k.startswith("vision_model.") for k in sd) else ""      # transformers 4.x vs 5.x
    vm = hf.vision_model if pre else hf
    sd[pre + "embeddings.patch_embedding.weight"] = p["conv_w"]
    sd[pre + "embeddings.patch_embedding.bias"] = p["conv_b"]
    sd[pre + "embeddings.position_embedding.weight"] = p["pos"]
    for i, lp in enumerate(p["layers"]):
        lpre = f"{pre}encoder.layers.{i}."
        w = lp["qkv_w"].view(16, 3, 72, 1152)
        b = lp["qkv_b"].view(16, 3, 72)
        for j, name in enumerate(("q_proj", "k_proj", "v_proj")):
            sd[lpre + f"self_attn.{name}.weight"] = w[:, j].reshape(1152, 1152)
            sd[lpre + f"self_attn.{name}.bias"] = b[:, j].reshape(1152)
        sd[lpre + "self_attn.out_proj.weight"], sd[lpre + "self_attn.out_proj.bias"] = lp["proj_w"], lp["proj_b"]
        sd[lpre + "layer_norm1.weight"], sd[lpre + "layer_norm1.bias"] = lp["ln1_w"], lp["ln1_b"]
        sd[lpre + "layer_norm2.weight"], sd[lpre + "layer_norm2.bias"] = lp["ln2_w"], lp["ln2_b"]
        sd[lpre + "mlp.fc1.weight"], sd[lpre + "mlp.fc1.bias"] = lp["fc1_w"], lp["fc1_b"]
        sd[lpre + "mlp.fc2.weight"], sd[lpre + "mlp.fc2.bias"] = lp["fc2_w"], lp["fc2_b"]
    hf.load_state_dict(sd)
    images = torch.randn(2, 3, 56, 56, generator=torch.Generator().manual_seed(1))
    with torch.no_grad():
        ref = vm.encoder(inputs_embeds=vm.embeddings(images)).last_hidden_state
    x = ovit.vit_embed(images, p, cfg)
    for lp in p["layers"]:
        x = ovit.vit_layer(x, lp, cfg)
    torch.testing.assert_close(x, ref, rtol=2e-4, atol=2e-4)


def test_get_external_inputs_matches_reference_fixture():
    """Token surgery for <image> / <video> tags, context-token indices, padding to 64: equal, integer for integer, to what the
    reference's own get_external_inputs produced under the same fake tokenizer / processor (oracle/make_golden.py)."""
    import types as _t

    from conftest import load_golden
    from long_vita_amd.inference_module import get_external_inputs
    g = load_golden("external_inputs.pt")

    class Tok:
        pad_token_id, eos_token_id = None, 151645

        def __call__(self, text, add_special_tokens=False):
            return _t.SimpleNamespace(input_ids=[g["table"][text]])

    class Proc:
        patch_size = 448

        def process_images_with_subpatch(self, spec):
            n = spec[0] * spec[1]
            return torch.zeros(n + 1 if n > 1 else 1, 3, 2, 2), (spec[0] * 448, spec[1] * 448)

        def process_images(self, lst):
            return torch.zeros(len(lst), 3, 2, 2)

        def process_video(self, n_frames, max_num_frame, max_fps):
            return torch.zeros(min(n_frames, max_num_frame), 3, 2, 2), None

    assert len(g["cases"]) == 5
    for c in g["cases"]:
        ext, toks, lens = get_external_inputs(torch.tensor(c["tokens"]), c.get("image_list"), c.get("image_path_list"),
                                              c.get("video_path_list"), Tok(), Proc(), image_token_length=256, max_num_frame=5,
                                              max_fps=1, device="cpu")
        assert torch.equal(toks, c["out_tokens"]) and torch.equal(lens, c["out_lengths"])
        assert torch.equal(ext["indices"], c["indices"]) and ext["indices"].dtype == torch.int64
        assert ext["images"].shape[0] == c["n_images"] and str(ext["images"].dtype) == c["images_dtype"]
        assert toks.shape[1] % 64 == 0


@pytest.mark.parametrize("cp", [2, 4, 8])
def test_per_rank_frame_loading_equals_reference_selection(cp):
    """get_external_inputs(cp_size=, cp_rank=) touches only the frames with a token on the rank, and what it returns is
    what the reference's selection (oracle.glue.get_batch_on_this_cp_rank, pinned by cp_batch.pt) keeps of the full
    output — images, and src / tgt indices after running the selection on it again (idempotence)."""
    import types as _t

    from oracle import glue
    from long_vita_amd.inference_module import frames_on_this_cp_rank, get_external_inputs
    IMG, VID = 900001, 900002
    table = {"<image>": IMG, "<video>": VID, "<IMG_CONTEXT>": 900003, "<img>": 900004, "</img>": 900005,
             "<VID_CONTEXT>": 900006, "<vid>": 900007, "</vid>": 900008, "<PATCH_CONTEXT>": 900009, "<patch>": 900010,
             "</patch>": 900011, "\n": 198}

    class Tok:
        pad_token_id, eos_token_id = None, 151645

        def __call__(self, text, add_special_tokens=False):
            return _t.SimpleNamespace(input_ids=[table[text]])

    class Proc:
        patch_size, image_size = 448, 2
        touched = []

        def process_images_with_subpatch(self, spec):
            n = spec[0] * spec[1]
            n = n + 1 if n > 1 else 1
            return torch.arange(n).view(n, 1, 1, 1).expand(n, 3, 2, 2).float() + 1000 * spec[2], (spec[0] * 448, spec[1] * 448)

        def process_images(self, lst):                       # a "frame" is its own id
            Proc.touched += [int(x[0, 0, 0]) for x in lst]
            return torch.stack([torch.full((3, 2, 2), float(x[0, 0, 0])) for x in lst])

    L = 8                                                    # context tokens per frame (the rule is length-agnostic)
    g = torch.Generator().manual_seed(cp)
    text = lambda n: torch.randint(0, 150000, (n,), generator=g).tolist()
    n_frames = 37
    video = [torch.full((1, 1, 3), 5000 + i) for i in range(n_frames)]
    # the reference reads image_list inside the <video> branch too (module.py:640-645), so a request carries one kind
    requests = [(text(11) + [VID] + text(29), None, dict(video_frames_list=[video]), n_frames),
                (text(5) + [IMG] + text(40) + [IMG] + text(17) + [IMG] + text(3), [(2, 1, 7), (1, 1, 8), (2, 3, 9)], {}, 3 + 1 + 7)]
    for row, image_list, extra, n_units in requests:
        kw = dict(image_token_length=L, max_num_frame=64, max_fps=1, device="cpu", **extra)
        full, toks, lens = get_external_inputs(torch.tensor([row]), image_list, None, None, Tok(), Proc(), **kw)
        S = toks.shape[1]
        assert S % (2 * cp) == 0 and full["images"].shape[0] == n_units
        pos = torch.arange(S).unsqueeze(0)
        n_empty = 0
        for r in range(cp):
            Proc.touched = []
            mine, toks_r, lens_r = get_external_inputs(torch.tensor([row]), image_list, None, None, Tok(), Proc(), cp_size=cp,
                                                       cp_rank=r, **kw)
            assert torch.equal(toks_r, toks) and torch.equal(lens_r, lens)
            keep = frames_on_this_cp_rank(full["indices"][1, :, 0].tolist(), L, S, cp, r)
            if not any(keep):
                assert mine["images"].shape[0] == 0 and mine["indices"].shape == (2, 0, L)
                n_empty += 1
                continue
            ref = glue.get_batch_on_this_cp_rank({"tokens": toks, "position_ids": pos, "external_images": full["images"].float(),
                                                  "external_indices": full["indices"]}, S, cp, r)
            assert torch.equal(mine["images"].float(), ref["external_images"])
            assert torch.equal(mine["indices"], full["indices"][:, torch.tensor(keep)])
            again = glue.get_batch_on_this_cp_rank({"tokens": toks, "position_ids": pos, "external_images": mine["images"].float(),
                                                    "external_indices": mine["indices"]}, S, cp, r)
            for k in ("external_images", "external_src_indices", "external_tgt_indices"):
                assert torch.equal(again[k], ref[k]), k
            if image_list is None:                       # only the rank's video frames went through the pixel path
                assert sorted(Proc.touched) == [5000 + i for i in range(n_frames) if keep[i]]
                assert len(Proc.touched) < n_frames
        assert n_empty < cp
    with pytest.raises(ValueError):
        get_external_inputs(torch.tensor([row]), image_list, None, None, Tok(), Proc(), cp_size=cp, cp_rank=cp, **kw)


def test_request_to_token_stream_chain_on_the_host():
    """inference_module.generate: prompt ids -> padded buffer (tokenization.py:150-166) -> tag expansion on the padded row ->
    true context length = expanded length - padding (module.py:357) -> decode loop.  With a position-revealing fake model
    the generated tokens are the true continuation of the EXPANDED prompt, and generation starts right behind it; the
    cached-path chunk length reaches the per-rank loader as a function of the expanded length."""
    import types as _t

    from long_vita_amd import generation, inference_module as im
    V = 900100
    table = {im.IMG_TAG_TOKEN: 900001, im.VID_TAG_TOKEN: 900002, im.IMG_CONTEXT_TOKEN: 900003, im.IMG_START_TOKEN: 900004,
             im.IMG_END_TOKEN: 900005, im.VID_CONTEXT_TOKEN: 900006, im.VID_START_TOKEN: 900007, im.VID_END_TOKEN: 900008,
             im.PATCH_CONTEXT_TOKEN: 900009, im.PATCH_START_TOKEN: 900010, im.PATCH_END_TOKEN: 900011, "\n": 198}

    class Tok:
        pad_token_id, eos_token_id = 0, 77

        def __call__(self, text, add_special_tokens=False):
            return _t.SimpleNamespace(input_ids=[table[text]])

    class Proc:
        patch_size, image_size = 448, 2
        seen = []

        def process_images(self, lst):
            Proc.seen.append(len(lst))
            return torch.zeros(len(lst), 3, 2, 2)

    nxt = lambda tok, pos: (tok * 7 + pos * 3 + 1) % 1000

    def model(tokens, position_ids, attention_mask, inference_params=None):
        sel = inference_params.logit_mask[0].nonzero().flatten()
        return torch.nn.functional.one_hot(nxt(tokens[0, sel], position_ids[0, sel]), 1000).float()[None]

    L, frames, gen_n = 4, 5, 9
    prompt = [11, 12, table[im.VID_TAG_TOKEN], 13, 14, 15]
    video = [torch.zeros(1, 1, 3) for _ in range(frames)]
    kw = dict(video_frames_list=[video], image_token_length=L, max_num_frame=16, device="cpu")
    tokens, lengths, ext = im.request_tensors(prompt, gen_n, Tok(), Proc(), **kw)
    expanded = len(prompt) - 1 + frames * (L + 2)
    assert int(lengths[0]) == expanded and tokens.shape[1] % 64 == 0 and tokens.shape[1] >= expanded + gen_n
    assert ext["indices"].shape == (2, frames, L)
    # the generation room is pad_token_id; the 64-padding behind it is eos because pad_token_id is falsy (module.py:684)
    assert tokens[0, expanded:expanded + gen_n].eq(0).all() and tokens[0, expanded + gen_n:].eq(77).all()
    out = list(im.generate(model, prompt, gen_n, Tok(), Proc(), use_kv_cache=False, termination_id=-1, **kw))
    final = out[-1][0][0]
    assert final.shape[0] == tokens.shape[1]                                  # runs to the end of the buffer without an eos
    assert torch.equal(final[:expanded], tokens[0, :expanded])
    for ctx in range(expanded, final.shape[0]):
        assert int(final[ctx]) == int(nxt(final[ctx - 1], torch.tensor(ctx - 1)))
    # text only: no padding to 64, no external inputs (module.py:321-334)
    t2, l2, e2 = im.request_tensors([5, 6, 7], 4, Tok(), None, device="cpu")
    assert t2.tolist() == [[5, 6, 7, 0, 0, 0, 0]] and l2.tolist() == [3] and e2 is None
    t3, _, _ = im.request_tensors([5, 6, 7], 0, Tok(), None, max_generate_length=10, device="cpu")
    assert t3.shape == (1, 10)
    # per-rank loading on the cached path: the chunk length is a function of the expanded prompt, not of the buffer
    cp = 2
    want_S = generation._cp_prefill_length(expanded, cp)
    seen = []
    im.request_tensors(prompt, gen_n, Tok(), Proc(), cp_size=cp, cp_rank=1,
                       cp_seq_length=lambda n: seen.append(n) or generation._cp_prefill_length(n - gen_n, cp), **kw)
    assert seen == [expanded + gen_n] and want_S == 1024


def test_c_abi_compiles_and_validates_from_plain_c(tmp_path):
    """include/vita_hip.h is C (gcc -std=c11 -Wall -Werror), the library resolves from C, argument validation answers without
    a GPU, and the ctypes struct mirrors have the sizes the C compiler gives the structs."""
    import ctypes

    from long_vita_amd import lib
    lib.build()
    exe = tmp_path / "abi_smoke"
    src = os.path.join(ROOT, "tests", "c", "abi_smoke.c")
    subprocess.run(["gcc", "-std=c11", "-Wall", "-Werror", "-I", os.path.join(ROOT, "include"), src, "-o", str(exe), "-ldl"],
                   check=True)
    out = subprocess.run([str(exe), lib.LIB_PATH], capture_output=True, text=True)
    assert out.returncode == 0 and "C ABI OK" in out.stdout, out.stdout + out.stderr
    sizes = dict(re.findall(r"sizeof\((\w+)\) = (\d+)", out.stdout))
    assert int(sizes["vita_attn_params"]) == ctypes.sizeof(lib.AttnParams)
    assert int(sizes["vita_decode_layer_params"]) == ctypes.sizeof(lib.DecodeLayerParams)


def test_adaptor_stacks_on_top_of_the_references_adaptor():
    """INTEGRATION.md §1's order — `import long_vita_megatron.megatron_adaptor` (the reference patches Megatron through ITS manager),
    then `import long_vita_amd.megatron_adaptor` (this package patches the same dotted names through its own) — on the stand-in
    Megatron tree (VERDICT r2 "missing" 5: two managers on the same names had no test).  The first manager is the reference's own
    class when /root/reference is readable (M/patch_utils.py is stdlib-only, loaded from its file), else this package's class under
    a second module name (pinned equal by patch_manager.pt); its replacements stand in for the reference's (same kinds: a `*_wrapper`
    for DotProductAttention.forward, outright replacements elsewhere, M/megatron_adaptor.py:21-22,81-106).  After the second
    adaptor every patched name resolves to THIS package's object — also in a module that imported the reference's replacement by
    name beforehand (identity propagation) — and the attention wrapper stacked on the reference's wrapper still routes to HIP."""
    import importlib.util
    import types
    import dummy_megatron as dm
    import long_vita_amd.megatron_adaptor as ad
    from long_vita_amd import layers
    from long_vita_amd.dot_product_attention import HipDotProductAttention
    from long_vita_amd.language_model_embedding import LanguageModelEmbedding
    from long_vita_amd.patch_utils import MindSpeedPatchesManager as aspm
    ref_file = "/root/reference/long_vita_megatron/patch_utils.py"
    src = ref_file if os.path.exists(ref_file) else os.path.join(ROOT, "long_vita_amd", "patch_utils.py")
    spec = importlib.util.spec_from_file_location("_first_manager_patch_utils", src)
    first = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(first)
    ref_mgr = first.MindSpeedPatchesManager
    ref_mgr.patches_info = {}
    aspm.patches_info = {}
    names = dm.install()
    try:
        calls = []

        def dot_product_attention_forward_wrapper(fn):                # the reference's kind of replacement: wraps Megatron's forward
            def wrapper(self, *a, **k):
                calls.append("reference wrapper")
                return fn(self, *a, **k)
            return wrapper

        class RefEmbedding:                                           # outright replacements (the reference's own classes)
            pass

        class RefColumnParallelLinear:
            pass

        def ref_local_spec(*a, **k):
            return "reference local spec"

        def ref_te_spec(*a, **k):
            return "reference te spec"

        for name, obj in [("megatron.core.transformer.dot_product_attention.DotProductAttention.forward", dot_product_attention_forward_wrapper),
                          ("megatron.core.models.gpt.gpt_layer_specs.get_gpt_layer_local_spec", ref_local_spec),
                          ("megatron.core.models.gpt.gpt_layer_specs.get_gpt_layer_with_transformer_engine_spec", ref_te_spec),
                          ("megatron.core.models.common.embeddings.language_model_embedding.LanguageModelEmbedding", RefEmbedding),
                          ("megatron.core.tensor_parallel.layers.ColumnParallelLinear", RefColumnParallelLinear)]:
            ref_mgr.register_patch(name, obj, create_dummy=True)
        # the reference's manager probes EVERY loaded module with hasattr(module, name) (M/patch_utils.py:67-70); a lazily-importing
        # package an earlier test loaded (transformers: its stubs import torchvision on first touch) answers that with an
        # ImportError, which has nothing to do with Megatron — keep such packages out of its sight while it runs
        lazy = {k: sys.modules.pop(k) for k in list(sys.modules) if k.split(".")[0] in ("transformers", "datasets", "accelerate")}
        try:
            ref_mgr.apply_patches()
        finally:
            sys.modules.update(lazy)
        specs = sys.modules["megatron.core.models.gpt.gpt_layer_specs"]
        assert specs.get_gpt_layer_local_spec() == "reference local spec"
        # a module of the host framework that imported the reference's replacements by name (pretrain_long_vita.py does)
        user = types.ModuleType("_host_entry_point")
        user.get_gpt_layer_with_transformer_engine_spec = specs.get_gpt_layer_with_transformer_engine_spec
        user.LanguageModelEmbedding = sys.modules["megatron.core.models.common.embeddings.language_model_embedding"].LanguageModelEmbedding
        sys.modules[user.__name__] = user
        assert user.LanguageModelEmbedding is RefEmbedding

        assert ad.exe_adaptation(create_dummy=True)                   # the second adaptor, on top
        assert sys.modules["megatron.core.models.common.embeddings.language_model_embedding"].LanguageModelEmbedding is LanguageModelEmbedding
        assert sys.modules["megatron.core.tensor_parallel.layers"].ColumnParallelLinear is layers.ColumnParallelLinear
        assert user.LanguageModelEmbedding is LanguageModelEmbedding                                    # propagated by identity
        assert user.get_gpt_layer_with_transformer_engine_spec is specs.get_gpt_layer_with_transformer_engine_spec
        te = dm.build_module(user.get_gpt_layer_with_transformer_engine_spec(), config=dm.TransformerConfig(use_cpu_initialization=True),
                             layer_number=1)
        assert isinstance(te.self_attention.core_attention, HipDotProductAttention) and isinstance(te.mlp, layers.GatedMLP)
        # DotProductAttention.forward: this package's wrapper wraps the reference-wrapped function and never calls it
        dpa = sys.modules["megatron.core.transformer.dot_product_attention"].DotProductAttention
        me = types.SimpleNamespace(num_attention_heads_per_partition=4, num_query_groups_per_partition=2, hidden_size_per_attention_head=128,
                                   attn_mask_type="AttnMaskType.causal")
        q = torch.zeros(8, 1, 4, 128, dtype=torch.bfloat16)
        with pytest.raises(RuntimeError, match="no CPU fallback"):    # reaches the HIP op (which refuses host tensors), not the reference's path
            dpa.forward(me, q, q[:, :, :2], q[:, :, :2], None, None, None)
        assert calls == []
    finally:
        sys.modules.pop("_host_entry_point", None)
        dm.uninstall(names)
        aspm.patches_info = {}
        ref_mgr.patches_info = {}


def test_cp_kv_split_keeps_every_attention_launch_at_ten_rounds_of_workgroups(monkeypatch):
    """ops.cp_kv_split: messages (= attention launches) per layer under CP.  40 : 8 heads — 128K at CP = 8 (S_l = 16384: 2560 workgroups
    per layer) takes ONE message, CP = 4 two, CP = 2 and the 1M prefill four; config 5's tensor-parallel half (20 : 4 heads, S_l = 32768)
    one; sizes too small for a single full launch keep the finest split; the override means "at most"."""
    from long_vita_amd import ops
    monkeypatch.delenv("VITA_CP_KV_SPLIT", raising=False)
    assert [ops.cp_kv_split(8, 40, s) for s in (16384, 32768, 65536, 131072)] == [1, 2, 4, 4]
    assert ops.cp_kv_split(4, 20, 32768) == 1 and ops.cp_kv_split(4, 20, 65536) == 2
    assert ops.cp_kv_split(8, 40, 2048) == 4 and ops.cp_kv_split(2, 4, 512) == 2 and ops.cp_kv_split(1, 5, 1024) == 1
    monkeypatch.setenv("VITA_CP_KV_SPLIT", "2")
    assert ops.cp_kv_split(8, 40, 16384) == 2
    assert ops.cp_kv_split(1, 5, 16384) == 1                       # at most: a tensor-parallel rank with one kv head
    monkeypatch.setenv("VITA_CP_KV_SPLIT", "3")
    assert ops.cp_kv_split(8, 40, 16384) == 2
    monkeypatch.setenv("VITA_CP_KV_SPLIT", "0")
    with pytest.raises(ValueError):
        ops.cp_kv_split(8, 40, 16384)


def test_vit_layer_specs_construct_on_cpu_with_the_references_parameter_names():
    """The three ViT spec builders registered on `long_vita_megatron.core.models.vision.vit_layer_specs.*` (M/core/models/vision/vit_layer_specs.py:30-101),
    built by the stand-in `build_module` on the CPU: layer classes, leaves, the parameter names the reference's checkpoints use (ls1 / ls2 only for
    InternViT), Megatron's shapes at SigLIP-400M's sizes (nothing padded), the activation / bias-chain flags — and the head-size padding helper."""
    import functools
    import dummy_megatron as dm
    import long_vita_amd.megatron_adaptor as ad
    from long_vita_amd import layers
    from long_vita_amd.dot_product_attention import HipDotProductAttention, _pad_head_dim
    from long_vita_amd.patch_utils import MindSpeedPatchesManager as aspm
    aspm.patches_info = {}
    names = dm.install()
    try:
        assert ad.exe_adaptation(create_dummy=True)
        vls = sys.modules["long_vita_megatron.core.models.vision.vit_layer_specs"]
        base = dict(num_query_groups=16, num_attention_heads=16, normalization="LayerNorm", add_bias_linear=True, add_qkv_bias=True,
                    gated_linear_unit=False, use_cpu_initialization=True)
        icfg = dm.TransformerConfig(hidden_size=1024, kv_channels=64, ffn_hidden_size=4096, activation_func=torch.nn.functional.gelu, **base)
        for builder, fused in ((vls.get_vit_layer_local_spec_for_intern, False), (vls.get_vit_layer_with_transformer_engine_spec_for_intern, True)):
            lyr = dm.build_module(builder(), config=icfg, layer_number=1)
            assert type(lyr).__name__ == "InternViTTransformerLayer" and isinstance(lyr.self_attention.core_attention, HipDotProductAttention)
            # local spec = Megatron's MLP with bias_activation_fusion off (M/pretrain_long_vita.py:213): fc1's bias meets the ROUNDED product;
            # TE spec = the bias inside the GEMM
            assert isinstance(lyr.mlp, layers.ViTMLP) and not lyr.mlp.tanh and lyr.mlp.unfused_bias == (not fused)
            keys = {k for k, _ in lyr.named_parameters()}
            assert {"ls1", "ls2", "mlp.linear_fc1.weight", "mlp.linear_fc2.bias", "self_attention.linear_proj.bias"} <= keys
            assert ("self_attention.linear_qkv.layer_norm_weight" in keys) == fused and ("input_layernorm.weight" in keys) == (not fused)
            assert float(lyr.ls1.detach().float().mean()) == pytest.approx(0.01, rel=1e-2)    # intern_vit_model.py:43-44
        scfg = dm.TransformerConfig(hidden_size=1152, kv_channels=72, ffn_hidden_size=4304,
                                    activation_func=functools.partial(torch.nn.functional.gelu, approximate="tanh"), **base)
        sl = dm.build_module(vls.get_vit_layer_local_spec_for_siglip(), config=scfg, layer_number=1)
        assert type(sl).__name__ == "SigLIPViTTransformerLayer" and sl.mlp.tanh and sl.mlp.unfused_bias
        shapes = {k: tuple(v.shape) for k, v in sl.named_parameters()}
        assert "ls1" not in shapes and shapes["mlp.linear_fc2.weight"] == (1152, 4304) and shapes["self_attention.linear_qkv.weight"] == (3456, 1152)
        bad = dm.TransformerConfig(hidden_size=1152, kv_channels=72, ffn_hidden_size=4304, activation_func=torch.nn.functional.silu, **base)
        with pytest.raises(NotImplementedError):
            dm.build_module(vls.get_vit_layer_local_spec_for_siglip(), config=bad, layer_number=1)
        # head sizes the kernels do not tile are zero-padded to the next one that is (scores and the kept output columns unchanged);
        # 64 / 96 / 128 pass through (r05: SigLIP's 72 -> 96; through r04 -> 128)
        t = torch.randn(2, 5, 3, 72)
        p_ = _pad_head_dim(t)
        assert p_.shape == (2, 5, 3, 96) and torch.equal(p_[..., :72], t) and float(p_[..., 72:].abs().max()) == 0.0
        assert _pad_head_dim(torch.zeros(1, 1, 1, 64)).shape[-1] == 64 and _pad_head_dim(torch.zeros(1, 1, 1, 40)).shape[-1] == 64
        assert _pad_head_dim(torch.zeros(1, 1, 1, 96)).shape[-1] == 96 and _pad_head_dim(torch.zeros(1, 1, 1, 100)).shape[-1] == 128
        with pytest.raises(NotImplementedError):
            _pad_head_dim(torch.zeros(1, 1, 1, 160))
    finally:
        dm.uninstall(names)
        aspm.patches_info = {}


def test_roctx_ranges_are_off_by_default_and_nest_under_vita_debug():
    """long_vita_amd/tracing.py: null contexts without VITA_DEBUG; with it, libroctx64's push / pop nest (return the depth) and every
    autograd Function's forward / backward is wrapped in a range (SURVEY.md §5 tracing)."""
    import subprocess, sys, textwrap
    from long_vita_amd import tracing
    if not tracing.ENABLED:
        import contextlib
        assert isinstance(tracing.range("x"), contextlib.nullcontext)
    code = textwrap.dedent("""
        import torch
        from long_vita_amd import tracing, autograd_fns
        assert tracing.ENABLED
        lib = tracing._load()
        if lib is None:
            print("NOLIB"); raise SystemExit(0)
        d0 = lib.roctxRangePushA(b"outer"); d1 = lib.roctxRangePushA(b"inner")
        assert d1 == d0 + 1, (d0, d1)
        assert lib.roctxRangePop() == d1 and lib.roctxRangePop() == d0
        with tracing.range("layer 0"):
            tracing.mark("m")
        assert autograd_fns.LinearFn.forward.__wrapped__ is not None and autograd_fns.FlashAttnFn.backward.__wrapped__ is not None
        print("OK")
    """)
    env = dict(os.environ, VITA_DEBUG="1")
    r = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, env=env, cwd=ROOT, timeout=300)
    assert r.returncode == 0, r.stderr[-2000:]
    assert "OK" in r.stdout or "NOLIB" in r.stdout


def test_tn_split_choice_never_leaves_an_empty_range_and_keeps_big_gradients_unsplit():
    """ops.tn_splits (host cost model of vita_gemm_bf16_tn_splitk): every range of the contraction is non-empty (the C side rejects an
    empty last range), the fp32 partials stay under 2 GiB, the decoder's wide gradients at 16K keep the one-pass kernel, the ViT's
    few-tile gradients and config 5's TP-halved qkv gradient are split."""
    from long_vita_amd import ops
    import itertools
    for M, N, K in itertools.product((256, 1024, 3584, 5120, 13824, 27648), (256, 1024, 4096, 5120), (512, 4096, 16384, 65600, 131072)):
        S = ops.tn_splits(M, N, K)
        nk = K // 64
        per = -(-nk // S)
        assert S >= 1 and nk - per * (S - 1) >= 1, (M, N, K, S)
        assert S == 1 or S * M * N * 4 <= (2 << 30), (M, N, K, S)
    assert ops.tn_splits(27648, 5120, 16384) == 1            # fc1 at 16K: 2160 tiles
    assert ops.tn_splits(13824, 5120, 16384) == 1
    assert ops.tn_splits(1024, 1024, 253 * 1025 // 64 * 64) > 1          # a ViT projection over 253 frames: 16 tiles
    assert ops.tn_splits(3584, 5120, 32768) > 1                # config 5: qkv at TP = 2


def test_committed_pmc_json_follows_from_the_committed_raw_counter_passes(tmp_path):
    """The profiles/rNN_attn128k_pmc.json that bench.py's `roofline.traffic` reads — the NEWEST committed one, bench.committed_pmc_sets()[0]
    — is tools/pmc_to_json.py applied to the committed raw rocprofv3 --pmc summaries and the committed kernel-trace statistics of the
    same round: re-deriving it here gives the same bytes per launch, and the corrected fetch figure is 2 x FETCH_SIZE KB
    (MI355X_MICROARCH.md's gfx950 correction) + WRITE_SIZE KB.  A round that commits a new json without its raw passes fails here."""
    import json
    import bench
    prof = os.path.join(ROOT, "profiles")
    newest = bench.committed_pmc_sets()[0]
    tag = os.path.basename(newest)[:3]
    assert int(tag[1:]) >= 5, newest
    raw, stats, committed = (os.path.join(prof, f) for f in (f"{tag}_attn128k_pmc_raw.txt", f"{tag}_bench128k_kernel_stats.txt", f"{tag}_attn128k_pmc.json"))
    for f in (raw, stats, os.path.join(prof, f"{tag}_bench128k_n1.json")):
        assert os.path.exists(f), f"{os.path.basename(newest)} is committed without {os.path.basename(f)}"
    out = tmp_path / "pmc.json"
    r = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "pmc_to_json.py"), raw, stats, str(out)], capture_output=True, text=True)
    assert r.returncode == 0, r.stderr[-1000:]
    a, b = json.load(open(out)), json.load(open(committed))
    assert a["hbm_bytes_per_launch"] == b["hbm_bytes_per_launch"]
    assert a["hbm_bytes_per_launch"] == a["FETCH_SIZE_KB"] * 1024 * 2 + a["WRITE_SIZE_KB"] * 1024
    assert a["algorithmic_bytes_per_launch"] == (2 * 131072 * 40 * 128 + 2 * 131072 * 8 * 128) * 2       # Q, O; K, V: bf16
    ms = a["sq_counters_S131072"]["ms_per_launch_rocprofv3_kernel_trace"]
    assert 100.0 < ms < 200.0
    # the bench line of the round quotes the same launch shape within a few percent of the kernel-trace average
    line = json.loads(open(os.path.join(prof, f"{tag}_bench128k_n1.json")).read().strip().splitlines()[-1])
    assert abs(line["roofline"]["ms_per_launch"] / ms - 1) < 0.03
    assert abs(line["roofline"]["frac"] - line["roofline"]["achieved"] / line["roofline"]["peak"]) < 1e-9


def test_header_is_self_contained_c11(tmp_path):
    """include/vita_hip.h compiles on its own as C11 (no prior #include in the translation unit) and as C++17."""
    hdr = os.path.join(ROOT, "include", "vita_hip.h")
    r = subprocess.run(["gcc", "-std=c11", "-Wall", "-Werror", "-fsyntax-only", "-x", "c", hdr], capture_output=True, text=True)
    assert r.returncode == 0, r.stderr[-2000:]
    r = subprocess.run(["g++", "-std=c++17", "-fsyntax-only", "-x", "c++", hdr], capture_output=True, text=True)
    assert r.returncode == 0, r.stderr[-2000:]


# ---- bench.py launches its own ranks -----------------------------------------------------------------------
def _bench_module():
    import importlib
    if ROOT not in sys.path:
        sys.path.insert(0, ROOT)
    return importlib.import_module("bench")


def test_bench_self_launch_command_is_the_drivers_form():
    """`python bench.py --gpus N` with no launcher re-runs itself as `python -m torch.distributed.run --nnodes=1 --nproc-per-node N
    --master-addr 127.0.0.1 --master-port P bench.py --gpus N ...` (the task statement's command for N > 1; the reference's scripts
    launch through torchrun themselves, R/scripts/megatron/qwen25/inference_qwen25_14b_intern_300m_server_cp.sh:96-181)."""
    bench = _bench_module()
    argv = ["--gpus", "8", "--steps", "20", "--warmup", "5"]
    cmd = bench.launch_command(8, 29511, argv)
    assert cmd[:3] == [sys.executable, "-m", "torch.distributed.run"]
    assert cmd[3:10] == ["--nnodes=1", "--nproc-per-node", "8", "--master-addr", "127.0.0.1", "--master-port", "29511"]
    assert os.path.samefile(cmd[10], os.path.join(ROOT, "bench.py")) and cmd[11:] == argv


def test_bench_self_launch_relays_one_line_and_degrades_instead_of_dying(capsys):
    """self_launch: a good first launch -> its line, rc 0, ONE launch; a launch that dies (or hangs past its limit) before a line ->
    exactly one more launch on the plain exchange schedule with the reason handed down, and THAT line comes out with rc 0; two bad
    launches -> non-zero.  The ranks never see VITA_BENCH_FORCE_SPAWN (they must not launch again) and always see dmabuf IPC."""
    bench = _bench_module()
    good = 'noise\n{"metric": "prefill tokens/sec/node (ViT+LLM) at seq=128K", "value": 1.0}\n'
    calls = []

    def runner(script):
        def run(cmd, env, timeout):
            calls.append((cmd, dict(env), timeout))
            return script[len(calls) - 1]
        return run

    env0 = {"PATH": os.environ.get("PATH", ""), "VITA_BENCH_FORCE_SPAWN": "1"}
    rc = bench.self_launch(2, ["--gpus", "2"], 2, 1, run=runner([(0, good)]), environ=env0)
    out = capsys.readouterr().out.strip().splitlines()
    assert rc == 0 and len(calls) == 1 and len(out) == 1 and out[0].startswith('{"metric"')
    cmd, env, limit = calls[0]
    assert "--nproc-per-node" in cmd and cmd[cmd.index("--nproc-per-node") + 1] == "2" and limit >= 900
    assert env["HSA_ENABLE_IPC_MODE_LEGACY"] == "0" and "VITA_BENCH_FORCE_SPAWN" not in env and "VITA_BENCH_DEGRADED" not in env
    assert env["VITA_BENCH_SELF_LAUNCHED"] == "1"

    for first in ((1, "Traceback ...\n"), (None, "")):                         # died / killed on the time limit
        calls.clear()
        rc = bench.self_launch(4, ["--gpus", "4"], 2, 1, run=runner([first, (0, good)]), environ=env0)
        out = capsys.readouterr().out.strip().splitlines()
        assert rc == 0 and len(calls) == 2 and len(out) == 1
        env2 = calls[1][1]
        assert all(env2[k] == v for k, v in bench.DEGRADED_ENV.items()) and "plain exchange schedule" in env2["VITA_BENCH_DEGRADED"]
        assert calls[0][0][calls[0][0].index("--master-port") + 1] != "" and "VITA_CP_STREAMS" not in calls[0][1]

    calls.clear()
    rc = bench.self_launch(2, ["--gpus", "2"], 2, 1, run=runner([(1, ""), (7, "")]), environ=env0)
    assert rc == 7 and len(calls) == 2 and capsys.readouterr().out.strip() == ""


def test_bench_run_ranks_ends_a_hung_launch_by_its_own_process_group():
    """_run_ranks: stdout captured, the return code passed on; a launch that outlives its limit is killed as a process group (its own
    session — nothing is matched by name) and reported as None."""
    bench = _bench_module()
    rc, out = bench._run_ranks([sys.executable, "-c", "print('{\"metric\": 1}')"], dict(os.environ), 60)
    assert rc == 0 and out.strip() == '{"metric": 1}'
    t0 = __import__("time").time()
    rc, out = bench._run_ranks([sys.executable, "-c", "import time, subprocess, sys; print('started', flush=True); "
                                "subprocess.Popen([sys.executable, '-c', 'import time; time.sleep(600)']); time.sleep(600)"], dict(os.environ), 3)
    assert rc is None and "started" in out and __import__("time").time() - t0 < 60


def test_padded_weight_is_never_stale_for_a_trainable_parameter():
    """optimizers write through `param.data`, which torch's version counter does not see — a trainable weight is padded
    per call (SigLIP's fc2, K = 4304), a frozen one is cached and follows `copy_` / `load_state_dict`; `invalidate_padded_weights`
    covers `.data` writes to frozen weights."""
    import torch
    from long_vita_amd import ops
    w = torch.nn.Parameter(torch.ones(4, 70))
    a = ops._padded_weight(w, 128)
    assert a.shape == (4, 128) and float(a[:, :70].min()) == 1.0 and float(a[:, 70:].abs().max()) == 0.0
    w.data.mul_(3.0)                                              # what Float16Optimizer._copy_main_params_to_model_params does
    assert float(ops._padded_weight(w, 128)[0, 0]) == 3.0
    f = torch.nn.Parameter(torch.ones(4, 70), requires_grad=False)
    b = ops._padded_weight(f, 128)
    assert ops._padded_weight(f, 128) is b                       # frozen: one copy
    with torch.no_grad():
        f.copy_(torch.full((4, 70), 2.0))                         # load_state_dict's write: bumps the version
    assert float(ops._padded_weight(f, 128)[0, 0]) == 2.0
    f.data.fill_(5.0)
    ops.invalidate_padded_weights()
    assert float(ops._padded_weight(f, 128)[0, 0]) == 5.0


def test_recompute_cache_marks_store_and_replay_runs_of_a_checkpointed_function(monkeypatch):
    """recompute_cache.checkpoint_wrapper around Megatron's tensor_parallel.checkpoint (restated in tests/dummy_megatron.py): the first run
    of the function is the "store" run, the run inside the backward the "replay" run; what the store run left comes back in call order,
    exactly once; outside a region, or with VITA_KEEP_ATTENTION unset, nothing is kept and the wrapper is a pass-through."""
    import torch
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import dummy_megatron as dm
    from long_vita_amd import recompute_cache as rc
    log = []

    def layer(x):
        region, phase = rc.current()
        log.append(phase)
        if phase == "store":
            rc.store(("a", 1)); rc.store(("b", 2))
        kept = [rc.take(), rc.take(), rc.take()] if phase == "replay" else None
        log.append(kept)
        return x * 2.0

    wrapped = rc.checkpoint_wrapper(dm.checkpoint)
    x = torch.ones(3, requires_grad=True)
    monkeypatch.delenv("VITA_KEEP_ATTENTION", raising=False)
    wrapped(layer, False, x).sum().backward()
    assert log == [None, None, None, None] and rc.current() == (None, None)            # pass-through: no region in either run
    log.clear()
    monkeypatch.setenv("VITA_KEEP_ATTENTION", "1")
    x.grad = None
    y = wrapped(layer, False, x)
    assert log == ["store", None] and rc.current() == (None, None)
    y.sum().backward()
    assert log[2] == "replay" and log[3] == [("a", 1), ("b", 2), None] and float(x.grad[0]) == 2.0
    assert rc.take() is None                                                            # outside any region
    # two regions alive at once (two checkpointed layers): each replay sees its own slots
    log.clear()
    x.grad = None
    def layer_n(tag):
        def f(t):
            _, phase = rc.current()
            if phase == "store":
                rc.store(tag)
            else:
                log.append((tag, rc.take()))
            return t + 1.0
        return f
    z = wrapped(layer_n("L1"), False, wrapped(layer_n("L0"), False, x))
    z.sum().backward()
    assert sorted(log) == [("L0", "L0"), ("L1", "L1")]


def test_recompute_cache_refuses_misaligned_replays(monkeypatch):
    """kept attention results are matched to replayed calls by call order.  A replay that leaves a kept result
    unconsumed raises (the store and replay runs made different calls); a kept context whose shape is not the replayed call's marks
    the region broken — that call and every later one recompute, nothing is raised."""
    import pytest
    import torch
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import dummy_megatron as dm
    from long_vita_amd import recompute_cache as rc
    monkeypatch.setenv("VITA_KEEP_ATTENTION", "1")
    wrapped = rc.checkpoint_wrapper(dm.checkpoint)

    def skips_one(x):
        _, phase = rc.current()
        if phase == "store":
            rc.store(("a", 1)); rc.store(("b", 2))
        else:
            rc.take()                                       # one of two
        return x * 2.0

    x = torch.ones(3, requires_grad=True)
    y = wrapped(skips_one, False, x)
    with pytest.raises(RuntimeError, match="were not consumed"):
        y.sum().backward()
    assert rc.current() == (None, None)

    got = []

    def wrong_shape(x):
        _, phase = rc.current()
        if phase == "store":
            rc.store((torch.zeros(1, 4, 2, 8), None)); rc.store((torch.zeros(1, 4, 2, 8), None))
        else:
            got.append(rc.take(like=torch.zeros(1, 6, 2, 8)))          # not the kept context's shape: broken from here on
            got.append(rc.take(like=torch.zeros(1, 4, 2, 8)))
        return x * 2.0

    x.grad = None
    wrapped(wrong_shape, False, x).sum().backward()
    assert got == [None, None] and float(x.grad[0]) == 2.0


def test_design_document_stays_reviewable():
    """VERDICT r04 housekeeping: DESIGN.md is the current-state document — at most 400 lines of at most 120 bytes (tools/wrap_md.py re-flows it);
    the round-by-round record lives in HISTORY.md."""
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    lines = open(os.path.join(root, "DESIGN.md"), "rb").read().split(b"\n")
    assert len(lines) <= 401, len(lines)                              # 400 lines + the empty piece after a final newline
    assert max(len(l) for l in lines) <= 120, max(len(l) for l in lines)
    assert os.path.exists(os.path.join(root, "HISTORY.md"))


def test_head_sizes_of_the_vit_attention_backward():
    """autograd_fns.non_causal_backward_plan: which head size the padded copies and the dQ pass of the ViT / SigLIP attention backward run at
    (r05: d = 96 instances; the pair kernel of attn_bwd_kvp.hip only exists at d = 128 and takes whole 256-row sequences)."""
    from long_vita_amd.autograd_fns import non_causal_backward_plan as plan
    assert plan(1025, 64) == (1152, 64, 64)                 # InternViT: native head size, both general kernels
    assert plan(1024, 96) == (1024, 128, 96)                # SigLIP at 448 px: copies at 128 for the pair kernel, dQ on d = 96 views
    assert plan(729, 96) == (768, 128, 96)
    assert plan(300, 96) == (384, 96, 96)                   # padded length not a multiple of 256: both general kernels at 96
    assert plan(1024, 96, "96") == (1024, 96, 96) and plan(1024, 96, "1") == (1024, 128, 128) and plan(1025, 64, "1") == (1152, 128, 128)
    assert plan(300, 128) == (384, 128, 128) and plan(100, 80)[1:] == (128, 128)       # anything else was padded to 128 by the caller
