"""`north_star`'s table on ONE GPU (VERDICT r3 "next round" 3): prefill seconds / tokens per second at S = 16K / 128K / 1M for
CP = 1 / 2 / 4 / 8.  CP = 1 cells are REAL single-GPU prefills (what `bench.py --seq S` runs).  CP > 1 cells are EMULATED: what ONE
rank of that CP run computes — its frames through the ViT, its two zig-zag chunks through the 48 layers with the attention running
against all S gathered keys through the chunk tables, the masked head, the logits exchange: the code path `bench.py --gpus CP` runs
on every rank — with the collectives replaced by local copies (every peer slot of the gathered K / V buffer receives this rank's own
shard: real values, so the kernels draw the power they would).  Communication time is NOT in an emulated cell: it is the compute side
of the scaling question (N-GPU prefill time >= max over ranks of the cell; bench.py's `comm` object reports the rest on real ranks).

    python tools/bench_table.py S:CP[:rank[:steps]] ...      e.g.  16384:1 16384:8:3 131072:4:1 1048576:8:3:1
Appends JSON lines to gpurun_out/r06_table.jsonl."""
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
LOG = open(os.path.join(ROOT, "gpurun_out", "r06_table.jsonl"), "a")
_real_all_gather = dist.all_gather_into_tensor
_CP = [1]


class _Group:
    pass


def fake_all_gather_into_tensor(out, inp, group=None, async_op=False):
    flat = out.view(_CP[0], -1)
    for q in range(_CP[0]):
        flat[q].copy_(inp.reshape(-1))
    return None


dist.all_gather_into_tensor = fake_all_gather_into_tensor
cfg, vcfg = gpt_vl_model.GPTConfig(), vision.VisionConfig()
vit = vision.MegatronVisionModel.random_init(vcfg, seed=4321, device=DEV)
model = gpt_vl_model.GPTVLModel.random_init(cfg, seed=1234, device=DEV, external_feature_model=vit)


def flops_per_token(seq, frames):
    sys.path.insert(0, ROOT)
    import bench
    return bench.flops_per_token(seq, frames, cfg, vcfg)


for spec in sys.argv[1:]:
    parts = [int(x) for x in spec.split(":")]
    seq, cp = parts[0], parts[1]
    rank = parts[2] if len(parts) > 2 else (cp // 2 if cp > 1 else 0)
    steps = parts[3] if len(parts) > 3 else (1 if seq >= 1 << 20 else 3)
    frames = synthetic.frames_for_seq(seq, tail_text=512)                  # what bench.py --seq S uses
    tokens, ext = synthetic.make_request(seq, frames, seed=1234, device=DEV)
    _CP[0] = cp
    mpu.set_context_parallel_state(cp, rank, _Group() if cp > 1 else None)
    model._ws = {}
    model.attn_events = None
    if seq < 1 << 20:                                                       # a 1M prefill takes minutes: no separate warm-up pass
        generation.prefill_step(model, tokens, seq, ext, reference_compat=False)
    else:                                                                   # ... but load every kernel first, at a short length
        t_s, e_s = synthetic.make_request(16384, synthetic.frames_for_seq(16384, tail_text=128), seed=1, device=DEV)
        generation.prefill_step(model, t_s, 16384, e_s, reference_compat=False)
        model._ws = {}
    torch.cuda.synchronize()
    torch.cuda.reset_peak_memory_stats()
    model.attn_events = []
    t0 = time.perf_counter()
    for _ in range(steps):
        out = generation.prefill_step(model, tokens, seq, ext, reference_compat=False)
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / steps
    ev = [a.elapsed_time(b) for a, b in model.attn_events]
    fpt = flops_per_token(seq, frames)
    s_l = seq // cp
    rec = dict(kind="table_cell", seq=seq, cp=cp, rank=rank, emulated=cp > 1, steps=steps, s_local=s_l, frames_total=frames,
               kv_messages_per_layer=ops.cp_kv_split(cfg.kv_groups, cfg.heads, s_l) if cp > 1 else 0,
               s_per_prefill=dt, tokens_per_s=seq / dt, attention_ms_per_layer=sum(ev) / max(len(ev), 1),
               algorithmic_gflop_per_token=fpt / 1e9, tflops_per_gpu=fpt * seq / cp / dt / 1e12,
               frac_of_mfma_peak=fpt * seq / cp / dt / 1e12 / 2500.0, peak_hbm_gb=torch.cuda.max_memory_allocated() / 2 ** 30,
               finite=bool(torch.isfinite(out.float()).all()),
               note=("one rank's compute, collectives replaced by local copies (no communication time)" if cp > 1 else "real single-GPU prefill"))
    print(json.dumps(rec), flush=True)
    LOG.write(json.dumps(rec) + "\n"); LOG.flush()
    del tokens, ext, out
    model._ws = {}
    torch.cuda.empty_cache()
