"""The TRAINING path Megatron would run, timed and sized (VERDICT r3 "missing" 2 / "next round" 1.ii).

`pretrain_long_vita.py` (M/pretrain_long_vita.py:841-869) drives torch autograd through the modules, with Megatron's block recompute
(`--recompute-granularity full --recompute-method block --recompute-num-layers N`, stage3 .sh:152-154) re-entering them through
`tensor_parallel.checkpoint`; every training number of rounds 1-3 came from training.TrainStep's hand-rolled sweep instead.  This
tool runs ONE full-width decoder layer (5120 / 40 : 8 / 13824) both ways on the same weights and input:

  module      the layer built by `build_module(get_gpt_layer_with_transformer_engine_spec(), config)` (tests/dummy_megatron.py stands in
              for Megatron-LM, which is not installable here): forward with autograd, `out.backward(go)`; kept = bytes alive after the forward
  module_ckpt the same inside tensor_parallel.checkpoint (a layer of the recompute block): forward under no_grad, backward = re-run + backward
  trainstep   TrainStep._layer_forward_keep / _layer_backward (a kept layer) and decoder_layer / _layer_backward(keep=None) (a recomputed one)

at  S = 16384 (TP = CP = 1)  and  config 5's per-rank geometry (TP = 2 x CP = 4: 32768 local rows against 131072 gathered keys, 20 : 4
heads, TP-halved GEMM widths, `--sequence-parallel`) with the collectives of the simulated rank replaced by local copies (one GPU).
Writes gpurun_out/r04_dropin_train.jsonl."""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch  # noqa: E402
import torch.distributed as dist  # noqa: E402

import dummy_megatron as dm  # noqa: E402
from long_vita_amd import gpt_vl_model, lib, parallel_state as mpu, tensor_parallel as tpar, training  # noqa: E402
from long_vita_amd.patch_utils import MindSpeedPatchesManager as aspm  # noqa: E402

lib.load(allow_build=False)
names = dm.install()
import long_vita_amd.megatron_adaptor as ad  # noqa: E402
aspm.patches_info = {}
assert ad.exe_adaptation(create_dummy=True)
specs = sys.modules["megatron.core.models.gpt.gpt_layer_specs"]
DEV = "cuda"
os.makedirs(os.path.join(ROOT, "gpurun_out"), exist_ok=True)
LOG = open(os.path.join(ROOT, "gpurun_out", "r04_dropin_train.jsonl"), "a")


def emit(**kw):
    s = json.dumps(kw)
    print(s, flush=True)
    LOG.write(s + "\n"); LOG.flush()


class _OneRank:
    """The collectives of ONE simulated rank of a (tp x cp) grid: gathers replicate the rank's own shard, reductions keep its part."""

    def __init__(self, n):
        self.n = n

    def size(self):
        return self.n


def _patch_collectives():
    def all_gather_into_tensor(out, inp, group=None, async_op=False):
        n = out.numel() // inp.numel()
        out.view(n, -1).copy_(inp.reshape(1, -1).expand(n, -1))

    def reduce_scatter_tensor(out, inp, group=None, async_op=False, op=None):
        n = inp.numel() // out.numel()
        out.view(-1).copy_(inp.view(n, -1)[0])

    def all_reduce(t, group=None, op=None, async_op=False):
        return None
    dist.all_gather_into_tensor, dist.reduce_scatter_tensor, dist.all_reduce = all_gather_into_tensor, reduce_scatter_tensor, all_reduce


def events(fn, n=3):
    fn(); torch.cuda.synchronize()
    ts = []
    for _ in range(n):
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record(); fn(); b.record(); torch.cuda.synchronize()
        ts.append(a.elapsed_time(b))
    return sorted(ts)[len(ts) // 2]


def run(tag, S_local, tp, cp, cp_rank=1):
    cfg = gpt_vl_model.GPTConfig(num_layers=1, vocab=1024)
    model = gpt_vl_model.GPTVLModel.random_init(cfg, seed=3, device=DEV)
    mpu.set_tensor_parallel_state(tp, 0, _OneRank(tp) if tp > 1 else None)
    mpu.set_context_parallel_state(cp, cp_rank, _OneRank(cp) if cp > 1 else None)
    if tp > 1:
        sharded, scfg = tpar.shard_llm_params(model.p, cfg, tp, 0)
        model = gpt_vl_model.GPTVLModel(scfg, sharded)
    lp, c = model.p["layers"][0], model.cfg
    mcfg = dm.TransformerConfig(hidden_size=cfg.hidden, num_attention_heads=cfg.heads, num_query_groups=cfg.kv_groups, kv_channels=cfg.head_dim,
                                ffn_hidden_size=cfg.ffn, sequence_parallel=tp > 1, tensor_model_parallel_size=tp, context_parallel_size=cp)
    layer = dm.build_module(specs.get_gpt_layer_with_transformer_engine_spec(), config=mcfg, layer_number=1)
    layer.load_state_dict({"self_attention.linear_qkv.weight": lp["qkv_w"], "self_attention.linear_qkv.bias": lp["qkv_b"],
                           "self_attention.linear_proj.weight": lp["o_w"], "mlp.linear_fc1.weight": lp["fc1_w"], "mlp.linear_fc2.weight": lp["fc2_w"],
                           "self_attention.linear_qkv.layer_norm_weight": lp["ln1"], "mlp.linear_fc1.layer_norm_weight": lp["ln2"]})
    layer.train()
    g = torch.Generator(device=DEV).manual_seed(S_local)
    rows_mod = S_local // tp                                   # sequence parallelism: the module sees its sequence shard
    x = (torch.randn(rows_mod, 1, cfg.hidden, generator=g, device=DEV) * 0.5).bfloat16()
    go = torch.randn(rows_mod, 1, cfg.hidden, generator=g, device=DEV).bfloat16()
    inv_freq = 1.0 / (cfg.rope_theta ** (torch.arange(0, cfg.head_dim, 2, dtype=torch.float32) / cfg.head_dim))
    ang = torch.outer(torch.arange(S_local, dtype=torch.float32), inv_freq)
    freqs = torch.cat((ang, ang), dim=-1)[:, None, None, :].to(DEV)
    MB = 2 ** 20

    # ---- the Megatron-built module under autograd --------------------------------------------------------------------------------------
    def mod_fwd(keep):
        xi = x.clone().requires_grad_(True)
        out = layer(xi, attention_mask=None, rotary_pos_emb=freqs)[0]
        keep.append((xi, out))

    def mod_step(ckpt):
        for q in layer.parameters():
            q.grad = None
        xi = x.clone().requires_grad_(True)
        if ckpt:
            out = dm.checkpoint(lambda t: layer(t, attention_mask=None, rotary_pos_emb=freqs)[0], False, xi)
        else:
            out = layer(xi, attention_mask=None, rotary_pos_emb=freqs)[0]
        out.backward(go)

    def fwd_only(ckpt):
        xi = x.clone().requires_grad_(True)
        if ckpt:
            return dm.checkpoint(lambda t: layer(t, attention_mask=None, rotary_pos_emb=freqs)[0], False, xi)
        return layer(xi, attention_mask=None, rotary_pos_emb=freqs)[0]

    res = {}
    for ckpt in (False, True):
        torch.cuda.synchronize(); torch.cuda.empty_cache()
        base = torch.cuda.memory_allocated()
        out = fwd_only(ckpt)
        torch.cuda.synchronize()
        kept = torch.cuda.memory_allocated() - base
        torch.cuda.reset_peak_memory_stats()
        out.backward(go)
        torch.cuda.synchronize()
        peak = torch.cuda.max_memory_allocated() - base
        del out
        for q in layer.parameters():
            q.grad = None
        t_f = events(lambda: fwd_only(ckpt))
        t_fb = events(lambda: mod_step(ckpt))
        res["module_ckpt" if ckpt else "module"] = dict(fwd_ms=t_f, fwd_bwd_ms=t_fb, bwd_ms=t_fb - t_f, kept_after_fwd_mb=kept / MB,
                                                        peak_in_bwd_mb=peak / MB)
    for q in layer.parameters():
        q.grad = None

    # ---- training.TrainStep's layer ------------------------------------------------------------------------------------------------------
    ts = training.TrainStep(model)
    h0 = (torch.randn(S_local, cfg.hidden, generator=g, device=DEV) * 0.5).bfloat16()     # the stand-alone driver runs TP ranks replicated
    dh0 = torch.randn(S_local, cfg.hidden, generator=g, device=DEV).bfloat16()
    cos, sin = model.rotary_pos_emb(S_local * cp)
    ws = model._workspace(S_local, h0.device)

    def ts_keep_fwd():
        return ts._layer_forward_keep(h0, lp, cos, sin)

    def ts_keep_step():
        out, keep = ts._layer_forward_keep(h0, lp, cos, sin)
        ts._layer_backward(dh0.clone(), h0, lp, cos, sin, {}, keep)

    def ts_rec_fwd():
        h = h0.clone()
        model.decoder_layer(h, lp, cos, sin, ws)
        return h

    def ts_rec_step():
        h = h0.clone()
        model.decoder_layer(h, lp, cos, sin, ws)
        ts._layer_backward(dh0.clone(), h0, lp, cos, sin, {}, None)

    with torch.no_grad():
        for name, f_fwd, f_step in (("trainstep_keep", ts_keep_fwd, ts_keep_step), ("trainstep_recompute", ts_rec_fwd, ts_rec_step)):
            torch.cuda.synchronize(); torch.cuda.empty_cache()
            base = torch.cuda.memory_allocated()
            kept_obj = f_fwd()
            torch.cuda.synchronize()
            kept = torch.cuda.memory_allocated() - base
            del kept_obj
            t_f = events(f_fwd)
            t_fb = events(f_step)
            res[name] = dict(fwd_ms=t_f, fwd_bwd_ms=t_fb, bwd_ms=t_fb - t_f, kept_after_fwd_mb=kept / MB)
    emit(kind="dropin_train_layer", geometry=tag, s_local=S_local, tp=tp, cp=cp, heads=f"{c.heads}:{c.kv_groups}",
         module_over_trainstep_kept=res["module"]["fwd_bwd_ms"] / res["trainstep_keep"]["fwd_bwd_ms"],
         module_ckpt_over_trainstep_recompute=res["module_ckpt"]["fwd_bwd_ms"] / res["trainstep_recompute"]["fwd_bwd_ms"], **res)
    model._ws = {}
    mpu.set_tensor_parallel_state(1, 0, None)
    mpu.set_context_parallel_state(1, 0, None)
    del layer, model, x, go, freqs, h0, dh0
    torch.cuda.empty_cache()


if __name__ == "__main__":
    which = sys.argv[1:] or ["16k", "cfg5"]
    _patch_collectives()
    with torch.autograd.set_multithreading_enabled(False):       # parallel_state is thread-local: keep the backward on this thread
        if "16k" in which:
            run("16K, TP = CP = 1", 16384, 1, 1, 0)
        if "cfg5" in which:
            run("config 5 per rank: TP 2 x CP 4, rank (tp 0, cp 1)", 32768, 2, 4, 1)
