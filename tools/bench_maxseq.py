"""BASELINE.json's "max seq fitting 8 x 288 GB", MEASURED instead of modelled (VERDICT r3 "next round" 4).

DESIGN.md 6.1 says one CP = 8 rank needs 35 GB + 14 KB * S for a prefill, so S ~ 17 M tokens fill 0.95 * 288 GB.  This tool emulates
ONE rank of a CP = 8 prefill at S = 16 777 216 tokens (S_l = 2 097 152 local rows, two zig-zag chunks of 1 048 576) on one MI355X:

  memory   EVERYTHING the rank holds is allocated for real — all 48 layers' weights + the ViT, the decoder workspace at S_l rows
           (h, x, qkv, ctx, SwiGLU act, K/V send buffer), the gathered K/V of a layer for all 16.8 M keys and 8 kv heads (68.7 GB),
           the rank's 8128 frames — and the allocator's peak is read back against the model;
  compute  the rank's frames through the whole ViT + projector, the visual scatter, then 2 of the 48 decoder layers at full width
           with the attention restricted to ONE kv group (5 of the 40 query heads; the other heads' context rows are zeroed): RMSNorm,
           QKV GEMM at M = 2 M rows, RoPE + K/V pack, the (emulated) all-gather into the rank-ordered buffer, the zig-zag chunk-table
           attention of 2 M query rows against 16.8 M keys, o-proj, MLP — then the final norm + masked head on 2 rows.  One kv group
           keeps the run to ~40 s per layer (the full layer is 8 x that: 4.5e16 flop per group and layer);
  parity   sampled query rows of that attention launch (all 5 heads) against an fp32 evaluation of the same rows over the SAME
           gathered buffer with plain torch ops on the device, chunked over the keys — the first run of every re-based buffer
           descriptor and int64 stride beyond 2^31 elements (the qkv buffer alone is 1.5e10 elements, the gathered K/V 3.4e10).
           (A checker on the device, not oracle/: 16.8 M keys x 128 x 2 tensors do not fit a CPU pass in the time budget.)

    python tools/bench_maxseq.py [S_l_log2=21] [layers=2]
Appends JSON lines to gpurun_out/r04_maxseq.jsonl."""
import json
import math
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402

from long_vita_amd import gpt_vl_model, lib, ops, parallel_state as mpu, vision  # noqa: E402

DEV = "cuda:0"
lib.load(allow_build=False)
os.makedirs(os.path.join(ROOT, "gpurun_out"), exist_ok=True)
LOG = open(os.path.join(ROOT, "gpurun_out", "r04_maxseq.jsonl"), "a")
GB = 2 ** 30


def emit(**kw):
    s = json.dumps(kw)
    print(s, flush=True)
    LOG.write(s + "\n"); LOG.flush()


@torch.no_grad()
def main():
    lg = int(sys.argv[1]) if len(sys.argv) > 1 else 21
    n_layers_run = int(sys.argv[2]) if len(sys.argv) > 2 else 2
    cp, rank = 8, 3
    s_l = 1 << lg
    S = s_l * cp
    c = s_l // 2
    cfg, vcfg = gpt_vl_model.GPTConfig(), vision.VisionConfig()
    torch.cuda.reset_peak_memory_stats()
    vit = vision.MegatronVisionModel.random_init(vcfg, seed=4321, device=DEV)
    model = gpt_vl_model.GPTVLModel.random_init(cfg, seed=1234, device=DEV, external_feature_model=vit)
    mpu.set_context_parallel_state(cp, rank, object())
    weights_gb = torch.cuda.memory_allocated() / GB
    emit(kind="maxseq_setup", S=S, s_local=s_l, cp=cp, rank=rank, weights_gb=weights_gb)

    # ---- the rank's frames and tokens: its two chunks hold whole frame blocks (258 tokens each) + a text tail -------------------------
    n_frames = s_l // 258
    g = torch.Generator(device=DEV).manual_seed(7)
    images = torch.empty(n_frames, 3, 448, 448, dtype=torch.bfloat16, device=DEV)
    for i in range(0, n_frames, 512):                                     # generated on the device, 512 frames at a time
        images[i:i + 512] = torch.randn(images[i:i + 512].shape, generator=g, device=DEV).bfloat16()
    tokens = torch.randint(0, 151643, (1, s_l), generator=g, device=DEV)
    fr = torch.arange(n_frames, device=DEV)
    col = torch.arange(256, device=DEV)
    tgt_s = (fr[:, None] * 258 + 1 + col[None, :]).reshape(-1)            # local positions of the context tokens
    ext = {"features": None, "src_indices": (fr[:, None].expand(-1, 256).reshape(-1), col[None, :].expand(n_frames, -1).reshape(-1)),
           "tgt_indices": (torch.zeros_like(tgt_s), tgt_s)}
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    feats = vit(images=images)                                            # 256-frame chunks (forward_chunk)
    torch.cuda.synchronize()
    t_vit = time.perf_counter() - t0
    ext["features"] = feats
    h = model.embedding(tokens, None, external_feature_dict=ext).view(s_l, cfg.hidden)
    ext.clear()                                                           # the projected features (20 GiB here) die with the scatter,
    del feats                                                             # as they do in GPTVLModel.forward
    emit(kind="maxseq_vit", frames=n_frames, seconds=t_vit, frames_per_s=n_frames / t_vit, after_embed_gb=torch.cuda.memory_allocated() / GB)

    # ---- the decoder workspace of the rank + the gathered K/V of one layer, as forward_cp holds them -----------------------------------
    ws = model._workspace(s_l, h.device)                                  # x, qkv, ctx, act, kv (n_msg, 2, s_l, 8 / n_msg, 128)
    n_msg = ws["kv"].shape[0]
    hg = cfg.kv_groups // n_msg
    gathered = torch.empty(n_msg, cp, 2 * s_l * hg * cfg.head_dim, dtype=torch.bfloat16, device=DEV)
    alloc_gb = torch.cuda.memory_allocated() / GB
    emit(kind="maxseq_alloc", allocated_gb=alloc_gb, workspace_kb_per_local_token=sum(v.numel() * 2 for v in ws.values()) / s_l / 1024 + 10,
         gathered_kv_gb=gathered.numel() * 2 / GB, kv_messages=n_msg,
         model_gb=(35e9 + 14e3 * S) / GB, model="DESIGN.md 6.1: 35 GB + 14 KB * S")
    # positions of the rank's rows: chunks `rank` and `2 cp - 1 - rank` of the global sequence
    pos = torch.cat([torch.arange(rank * c, (rank + 1) * c, device=DEV), torch.arange((2 * cp - 1 - rank) * c, (2 * cp - rank) * c, device=DEV)])
    cos, sin = ops.rope_table(pos, model.rotary_pos_emb.inv_freq)
    kv_gid, kv_row = [], []
    for p_ in range(cp):
        kv_gid += [p_, 2 * cp - 1 - p_]
        kv_row += [p_ * 2 * s_l, p_ * 2 * s_l + c]
    own = [rank, 2 * cp - 1 - rank]
    qpg, d = cfg.qpg, cfg.head_dim
    times = []
    checks = []
    for li in range(n_layers_run):
        lp = model.p["layers"][li]
        ev = [torch.cuda.Event(enable_timing=True) for _ in range(6)]
        ev[0].record()
        x = ops.rmsnorm(h, lp["ln1"], cfg.eps, out=ws["x"])
        qkv = ops.gemm(x, lp["qkv_w"], ops.EPI_BIAS, lp["qkv_b"], out=ws["qkv"])
        ops.rope_qkv_(qkv, cfg.kv_groups, qpg, d, cos, sin, ws["kv"], n_msg)
        ev[1].record()
        for j in range(n_msg):                                            # the all-gather, emulated: every peer slot <- this rank's shard
            gathered[j].copy_(ws["kv"][j].reshape(1, -1).expand(cp, -1))
        ev[2].record()
        m5 = qkv.view(1, s_l, cfg.kv_groups, qpg + 2, d)
        q1 = m5[:, :, 0:1, :qpg]                                          # ONE kv group: 5 query heads (grouped view, read in place)
        rows = gathered[0].view(cp * 2 * s_l, hg, d)                      # K rows of rank p at p * 2 s_l, V at + s_l
        ws["ctx"].zero_()
        o1 = ws["ctx"][:, :, :qpg]
        lse = torch.empty(1, qpg, s_l, dtype=torch.float32, device=DEV)
        ops.flash_attn(q1, rows[:, 0:1].unsqueeze(0), rows[s_l:, 0:1].unsqueeze(0), causal=True, chunk_len=c, q_chunk_gid=own,
                       kv_chunk_gid=kv_gid, kv_chunk_row=kv_row, out=o1, lse_out=lse)
        ev[3].record()
        if li == 0:
            checks = sampled_rows_check(q1, rows, o1, s_l, c, own, kv_gid, kv_row, qpg, d)
        ev[4].record()
        ops.gemm(ws["ctx"].view(s_l, cfg.heads * d), lp["o_w"], ops.EPI_RESIDUAL, residual=h, out=h)
        x = ops.rmsnorm(h, lp["ln2"], cfg.eps, out=ws["x"])
        act = ops.gemm(x, lp["fc1_w"], ops.EPI_SWIGLU, out=ws["act"])
        ops.gemm(act, lp["fc2_w"], ops.EPI_RESIDUAL, residual=h, out=h)
        ev[5].record()
        torch.cuda.synchronize()
        pairs = float(c) * c * (own[0] + 0.5) + float(c) * c * (own[1] + 0.5)
        fl = 4.0 * d * qpg * pairs
        t_attn = ev[2].elapsed_time(ev[3])
        times.append(dict(layer=li, norm_qkv_rope_ms=ev[0].elapsed_time(ev[1]), gather_copy_ms=ev[1].elapsed_time(ev[2]), attention_one_kv_group_ms=t_attn,
                          attention_tflops=fl / t_attn / 1e9, proj_mlp_ms=ev[4].elapsed_time(ev[5]), finite=bool(torch.isfinite(h[::65537].float()).all())))
        emit(kind="maxseq_layer", **times[-1])
    # ---- final norm + masked head on the two rows this rank marks ---------------------------------------------------------------------
    idx = torch.tensor([c - 1, s_l - 1], device=DEV)
    rows2 = ops.rmsnorm(ops.row_gather(h, idx), model.p["final_ln"], cfg.eps)
    logits = ops.gemm_skinny(rows2, model.p["lm_head"])
    torch.cuda.synchronize()
    peak = torch.cuda.max_memory_allocated() / GB
    free_b, total_b = torch.cuda.mem_get_info()
    fixed = weights_gb + 5.0                                             # weights + the ViT chunk buffers: what does not grow with S
    s_max = int(S * (0.95 * total_b / GB - fixed) / max(peak - fixed, 1e-9))
    emit(kind="maxseq_result", device_total_gb=total_b / GB, max_seq_at_95pct_by_the_measured_slope=s_max, S=S, s_local=s_l, cp=cp, rank=rank, frames_on_rank=n_frames, layers_run=n_layers_run,
         peak_hbm_gb=peak, model_gb=(35e9 + 14e3 * S) / GB, fits=peak < 0.95 * total_b / GB, logits_finite=bool(torch.isfinite(logits.float()).all()),
         attention_parity=checks,
         note="peak = every buffer one CP = 8 rank holds for a prefill at this S (48 layers of weights, workspace at S_l rows, gathered K/V of one layer "
              "for all 8 kv heads, the rank's frames, ViT chunk buffers); compute = ViT over the rank's frames + 2 decoder layers with one kv group's attention")


def sampled_rows_check(q1, rows, o1, s_l, c, own, kv_gid, kv_row, qpg, d):
    """fp32 attention of a few query rows (first / middle / last rows of both chunks, all 5 heads) over the same gathered buffer with torch ops."""
    scale = 1.0 / math.sqrt(d)
    out = []
    picks = [(0, 0), (0, 4097), (0, c // 2 + 13), (0, c - 1), (1, 0), (1, c // 3 + 5), (1, c - 1)]
    for qc, i in picks:
        r_loc = qc * c + i
        qv = q1[0, r_loc, 0].float()                                      # [qpg, d]
        m = torch.full((qpg,), -float("inf"), device=q1.device)
        l = torch.zeros(qpg, device=q1.device)
        acc = torch.zeros(qpg, d, device=q1.device)
        for j, (gid, row0) in enumerate(zip(kv_gid, kv_row)):
            if gid > own[qc]:
                continue
            n_vis = c if gid < own[qc] else i + 1
            for a in range(0, n_vis, 1 << 18):
                b = min(a + (1 << 18), n_vis)
                k = rows[row0 + a: row0 + b, 0].float()                   # [n, d]
                v = rows[s_l + row0 + a: s_l + row0 + b, 0].float()
                s = (qv @ k.t()) * scale                                  # [qpg, n]
                m_new = torch.maximum(m, s.max(dim=1).values)
                p = torch.exp(s - m_new[:, None])
                corr = torch.exp(m - m_new)
                l = l * corr + p.sum(dim=1)
                acc = acc * corr[:, None] + p @ v
                m = m_new
        want = acc / l[:, None]
        got = o1[0, r_loc].float()
        out.append(dict(chunk=qc, row=i, rel_l2=float((got - want).norm() / want.norm()), max_abs=float((got - want).abs().max())))
    return out


if __name__ == "__main__":
    main()
