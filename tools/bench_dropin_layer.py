"""The drop-in path timed (VERDICT r2 "next round" 8): one decoder layer at the 14B width built the way Megatron builds it —
`build_module(get_gpt_layer_with_transformer_engine_spec(), config=...)` through tests/dummy_megatron.py (Megatron-LM itself is not
installable here) — against GPTVLModel.decoder_layer, the fused stand-alone driver bench.py measures, on the same weights and input,
without autograd (the prefill).  Prints JSON lines; writes gpurun_out/r03_dropin_layer.jsonl.

What the module path cannot fuse (Megatron's TransformerLayer / SelfAttention wiring owns these steps): the residual adds behind
linear_proj / linear_fc2 (a bias-dropout-add call: one vita_add_bf16 pass each instead of a GEMM epilogue), RoPE as two calls on
separate q / k tensors, and the copies Megatron's SelfAttention makes when it splits the mixed QKV (`.contiguous()`)."""
import json, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch
import dummy_megatron as dm
from long_vita_amd import gpt_vl_model, lib
from long_vita_amd.patch_utils import MindSpeedPatchesManager as aspm
lib.load(allow_build=False)
names = dm.install()
import long_vita_amd.megatron_adaptor as ad
aspm.patches_info = {}
assert ad.exe_adaptation(create_dummy=True)
specs = sys.modules["megatron.core.models.gpt.gpt_layer_specs"]
DEV = "cuda"
os.makedirs(os.path.join(ROOT, "gpurun_out"), exist_ok=True)
LOG = open(os.path.join(ROOT, "gpurun_out", "r03_dropin_layer.jsonl"), "a")
cfg = gpt_vl_model.GPTConfig(num_layers=1, vocab=1024)
model = gpt_vl_model.GPTVLModel.random_init(cfg, seed=3, device=DEV)
lp = model.p["layers"][0]
mcfg = dm.TransformerConfig(hidden_size=cfg.hidden, num_attention_heads=cfg.heads, num_query_groups=cfg.kv_groups, kv_channels=cfg.head_dim,
                            ffn_hidden_size=cfg.ffn)
layer = dm.build_module(specs.get_gpt_layer_with_transformer_engine_spec(), config=mcfg, layer_number=1)
layer.load_state_dict({"self_attention.linear_qkv.weight": lp["qkv_w"], "self_attention.linear_qkv.bias": lp["qkv_b"],
                       "self_attention.linear_proj.weight": lp["o_w"], "mlp.linear_fc1.weight": lp["fc1_w"], "mlp.linear_fc2.weight": lp["fc2_w"],
                       "self_attention.linear_qkv.layer_norm_weight": lp["ln1"], "mlp.linear_fc1.layer_norm_weight": lp["ln2"]})
for S in [int(x) for x in (sys.argv[1:] or ["16384", "131072"])]:
    g = torch.Generator(device=DEV).manual_seed(S)
    x = (torch.randn(S, cfg.hidden, generator=g, device=DEV) * 0.5).bfloat16()
    # the fp32 angle table as Megatron's RotaryEmbedding hands it over: [s, 1, 1, d] = cat(outer(pos, inv_freq)) x 2 (rotary_pos_embedding.py:74-122)
    inv_freq = 1.0 / (cfg.rope_theta ** (torch.arange(0, cfg.head_dim, 2, dtype=torch.float32) / cfg.head_dim))
    ang = torch.outer(torch.arange(S, dtype=torch.float32), inv_freq)
    freqs = torch.cat((ang, ang), dim=-1)[:, None, None, :].to(DEV)
    cos, sin = model.rotary_pos_emb(S)
    ws = model._workspace(S, x.device)

    def fused():
        h = x.clone()
        return model.decoder_layer(h, lp, cos, sin, ws)

    def module():
        with torch.no_grad():
            return layer(x.view(S, 1, -1), attention_mask=None, rotary_pos_emb=freqs)[0]

    def timeit(fn, n=3):
        fn(); torch.cuda.synchronize()
        ts = []
        for _ in range(n):
            a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            a.record(); fn(); b.record(); torch.cuda.synchronize()
            ts.append(a.elapsed_time(b))
        return sorted(ts)[len(ts) // 2]

    o_f, o_m = fused(), module().view(S, -1)
    rel = float((o_m.float() - o_f.float()).norm() / o_f.float().norm())
    t_clone = timeit(lambda: x.clone())
    t_f, t_m = timeit(fused) - t_clone, timeit(module)
    rec = {"kind": "dropin_layer", "S": S, "fused_driver_ms": t_f, "megatron_built_module_ms": t_m, "module_over_fused": t_m / t_f,
           "rel_l2_module_vs_fused": rel}
    print(json.dumps(rec), flush=True)
    LOG.write(json.dumps(rec) + "\n"); LOG.flush()
    model._ws = {}
    del x, freqs, o_f, o_m
    torch.cuda.empty_cache()
