"""ViT layer forward + backward through the Megatron-built module (VERDICT r3 "next round" 6): the InternViT layer of
`get_vit_layer_local_spec_for_intern()` (tests/dummy_megatron.py stands in for Megatron-LM) at 253 / 64 frames x 1025 tokens, autograd on,
with the attention backward at the native head size 64 (r04: attn_bwd.hip templated on d) against the r03 path that zero-padded q / k / v / o /
dO to d = 128 (VITA_VIT_BWD_PAD128=1).  Appends JSON lines to gpurun_out/r05_vit_layer.jsonl."""
import json, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch
import dummy_megatron as dm
from long_vita_amd import lib
from long_vita_amd.patch_utils import MindSpeedPatchesManager as aspm
lib.load(allow_build=False)
names = dm.install()
import long_vita_amd.megatron_adaptor as ad
aspm.patches_info = {}
assert ad.exe_adaptation(create_dummy=True)
vls = sys.modules["long_vita_megatron.core.models.vision.vit_layer_specs"]
DEV = "cuda"
LOG = open(os.path.join(ROOT, "gpurun_out", "r05_vit_layer.jsonl"), "a")
mcfg = dm.TransformerConfig(hidden_size=1024, num_attention_heads=16, num_query_groups=16, kv_channels=64, ffn_hidden_size=4096,
                            normalization="LayerNorm", layernorm_epsilon=1e-6, add_bias_linear=True, add_qkv_bias=True, gated_linear_unit=False,
                            activation_func=torch.nn.functional.gelu)
layer = dm.build_module(vls.get_vit_layer_local_spec_for_intern(), config=mcfg, layer_number=1)
layer.train()


def events(fn, n=3):
    fn(); torch.cuda.synchronize()
    ts = []
    for _ in range(n):
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record(); fn(); b.record(); torch.cuda.synchronize()
        ts.append(a.elapsed_time(b))
    return sorted(ts)[len(ts) // 2]


for frames in [int(x) for x in (sys.argv[1:] or ["253", "64"])]:
    g = torch.Generator(device=DEV).manual_seed(frames)
    x = (torch.randn(1025, frames, 1024, generator=g, device=DEV) * 0.5).bfloat16()
    go = torch.randn(1025, frames, 1024, generator=g, device=DEV).bfloat16()

    def fwd():
        with torch.no_grad():
            return layer(x, attention_mask=None)[0]

    def step():
        for q in layer.parameters():
            q.grad = None
        xi = x.clone().requires_grad_(True)
        out = layer(xi, attention_mask=None)[0]
        out.backward(go)
        return xi.grad

    rec = {"kind": "vit_layer", "frames": frames, "fwd_inference_ms": events(fwd)}
    grads = {}
    for tag, env in (("native_d64", "0"), ("padded_d128_r03", "1")):
        os.environ["VITA_VIT_BWD_PAD128"] = env
        grads[tag] = step().clone()
        rec[tag + "_fwd_bwd_ms"] = events(step)
    os.environ["VITA_VIT_BWD_PAD128"] = "0"
    a, b = grads["native_d64"].float(), grads["padded_d128_r03"].float()
    rec["speedup"] = rec["padded_d128_r03_fwd_bwd_ms"] / rec["native_d64_fwd_bwd_ms"]
    rec["dx_rel_l2_native_vs_padded"] = float((a - b).norm() / b.norm())
    print(json.dumps(rec), flush=True)
    LOG.write(json.dumps(rec) + "\n"); LOG.flush()
