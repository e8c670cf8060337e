"""What ONE rank of the CP = 8 Long-VITA-128K prefill computes, measured on one GPU (no second GPU is available to the builder):
the rank's frames through the ViT, its two zig-zag chunks (S_l = 16384 tokens) through the 48 layers with the attention running
against all 131072 gathered keys through the CP = 8 chunk tables, the masked head and the logits exchange — the code path
`bench.py --gpus 8` runs on every rank, with the collectives replaced by local copies (every peer slot of the gathered K / V buffer
receives this rank's own shard: real values, so the kernels draw the power they would).  Communication time is NOT in it; what it
gives is the compute side of the scaling question: 8-GPU prefill time >= max over ranks of this number.

    python tools/bench_cp8_rank_prefill.py [rank ...]        (default ranks 0 3 7)
Prints JSON lines; appends to gpurun_out/r03_cp8_rank_prefill.jsonl."""
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402
import torch.distributed as dist  # noqa: E402

from long_vita_amd import generation, gpt_vl_model, lib, ops, parallel_state as mpu, synthetic, vision  # noqa: E402

DEV = "cuda:0"
lib.load(allow_build=False)
os.makedirs(os.path.join(ROOT, "gpurun_out"), exist_ok=True)
LOG = open(os.path.join(ROOT, "gpurun_out", "r03_cp8_rank_prefill.jsonl"), "a")
CP, SEQ = int(os.environ.get("CP", "8")), int(os.environ.get("SEQ", "131072"))


class _Group:
    pass


def fake_all_gather_into_tensor(out, inp, group=None, async_op=False):
    flat = out.view(CP, -1)
    for q in range(CP):
        flat[q].copy_(inp.reshape(-1))
    return None


dist.all_gather_into_tensor = fake_all_gather_into_tensor
cfg, vcfg = gpt_vl_model.GPTConfig(), vision.VisionConfig()
frames = synthetic.frames_for_seq(SEQ, tail_text=512)
vit = vision.MegatronVisionModel.random_init(vcfg, seed=4321, device=DEV)
model = gpt_vl_model.GPTVLModel.random_init(cfg, seed=1234, device=DEV, external_feature_model=vit)
tokens, ext = synthetic.make_request(SEQ, frames, seed=1234, device=DEV)
for r in [int(x) for x in (sys.argv[1:] or ["0", "3", "7"])]:
    mpu.set_context_parallel_state(CP, r, _Group())
    model._ws = {}
    model.attn_events = None
    generation.prefill_step(model, tokens, SEQ, ext, reference_compat=False)          # warm-up
    torch.cuda.synchronize()
    model.attn_events = []
    t0 = time.perf_counter()
    n = 3
    for _ in range(n):
        out = generation.prefill_step(model, tokens, SEQ, ext, reference_compat=False)
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / n
    ev = [a.elapsed_time(b) for a, b in model.attn_events]
    s_l = SEQ // CP
    rec = dict(kind="cp_rank_prefill", cp=CP, rank=r, seq=SEQ, s_local=s_l, frames_total=frames,
               kv_messages_per_layer=ops.cp_kv_split(cfg.kv_groups, cfg.heads, s_l), s_per_prefill_compute_only=dt,
               attention_ms_per_layer=sum(ev) / max(len(ev), 1), tokens_per_s_if_every_rank_took_this_long=SEQ / dt,
               finite=bool(torch.isfinite(out.float()).all()))
    print(json.dumps(rec), flush=True)
    LOG.write(json.dumps(rec) + "\n"); LOG.flush()
