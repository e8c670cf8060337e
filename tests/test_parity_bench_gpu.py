"""Parity at the BENCHMARK configurations (VERDICT r1 "next round" item 1), with the achieved errors recorded.

  * vita_flash_attn_fwd at S = 16384 and 131072 (40:8 heads, d = 128, causal) — BASELINE configs 2 / 3 on one GPU;
  * the CP = 8 chunk-table geometries of configs 3 and 4 for one rank and one kv-head split: S_l = 16384 local queries
    against 131072 gathered keys, S_l = 131072 against 1 M keys (the packed [rank][K | V][S_l][hg][d] buffer that
    dot_product_attention.forward_cp hands to the kernel);
  * one full-width (5120 / 40:8 / 13824) 48-layer prefill at 16K.
The oracle is evaluated on SAMPLED query rows only (first / last row of chunks and 256-row tiles, 64-key tile edges,
random rows): fp32 `oracle.attention.core_attention` on the CPU with the rows' global positions — restating
M/core/transformer/dot_product_attention.py:186-289 with the zig-zag ownership of M/training/utils.py:329-341.
Every achieved rel-L2 / max-abs lands in gpurun_out/r03_parity.json (copied to profiles/); each limit below is
<= 1.5 x the value measured on the MI355X.
"""
import json
import math
import os

import pytest
import torch

pytestmark = pytest.mark.gpu

from oracle import attention as oattn, glue, llm as ollm  # noqa: E402

DEV = "cuda"
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
from conftest import record_parity  # noqa: E402

# name -> (rel-L2 limit, max-abs limit) = 1.5 x the values measured on the MI355X (profiles/r02_parity.json):
#   attention, single chunk : rel-L2 1.18e-3, max-abs 9.3e-3 (outputs of rms 0.16: early rows dominate the norm)
#   attention, CP = 8 tables: rel-L2 2.31e-3, max-abs 1.5e-4 / 3.6e-5 (rows that average over 10^5..10^6 keys: rms 7.6e-3 / 2.4e-3)
# The attention error is the bf16 rounding of P before P V (as flash-attn / TE round it); north_star's 1e-3 is met to within 20 %.
LIMITS = {
    "attn_fwd_S16384": (1.8e-3, 1.4e-2),
    "attn_fwd_S131072": (1.8e-3, 1.4e-2),
    "attn_cp8_Sl16384_S131072": (3.5e-3, 2.2e-4),
    "attn_cp8_Sl131072_S1048576": (3.5e-3, 5.5e-5),
    # attention backward at 16K (dQ / dK / dV of kv groups 0 and 7 against fp32 autograd through the oracle on the same rows):
    # measured rel-L2 2.43e-3 / 2.42e-3 / 2.32e-3, max-abs 8.4e-3 / 1.6e-2 / 2.6e-2 (gradient rms 0.034 / 0.076 / 0.083)
    "attn_bwd_S16384_dq": (3.6e-3, 1.3e-2),
    "attn_bwd_S16384_dk": (3.6e-3, 2.4e-2),
    "attn_bwd_S16384_dv": (3.6e-3, 4.0e-2),
}


def record(name, **metrics):
    record_parity(name, **metrics)


def check(name, got, ref, **extra):
    got, ref = got.float().cpu(), ref.float().cpu()
    rel = float((got - ref).norm() / ref.norm())
    mab = float((got - ref).abs().max())
    record(name, rel_l2=rel, max_abs=mab, ref_rms=float(ref.pow(2).mean().sqrt()), limit_rel_l2=LIMITS[name][0],
           limit_max_abs=LIMITS[name][1], **extra)
    assert rel < LIMITS[name][0], (name, rel)
    assert mab < LIMITS[name][1], (name, mab)


@pytest.fixture(scope="module")
def ops():
    from long_vita_amd import ops as _ops
    _ops._L.load(allow_build=False)
    return _ops


def sample_rows(n_rows, chunk, n=64, seed=0):
    """First / last row of every chunk, 256-row tile edges and 64-key tile edges around them, then random rows."""
    rows = set()
    for c0 in range(0, n_rows, chunk):
        rows.update([c0, c0 + 1, c0 + 63, c0 + 64, c0 + 255, c0 + 256, c0 + chunk - 257, c0 + chunk - 256, c0 + chunk - 65,
                     c0 + chunk - 64, c0 + chunk - 1])
    rows = {r for r in rows if 0 <= r < n_rows}
    g = torch.Generator().manual_seed(seed)
    while len(rows) < n:
        rows.add(int(torch.randint(0, n_rows, (1,), generator=g)))
    return torch.tensor(sorted(rows), dtype=torch.int64)


def oracle_rows(q_rows, k, v, q_pos, group):
    """q_rows [n, G, d] (the G query heads of kv group `group`), k / v [S, ng, d] in global order -> [n, G, d] fp32, on the CPU."""
    n, G, d = q_rows.shape
    out = oattn.core_attention(q_rows.float().cpu().view(n, 1, G, d), k[:, group:group + 1].float().cpu().unsqueeze(1),
                               v[:, group:group + 1].float().cpu().unsqueeze(1), True, q_pos=q_pos.cpu())
    return out.view(n, G, d)


def randn_bf16(shape, seed):
    g = torch.Generator(device=DEV).manual_seed(seed)
    return torch.randn(*shape, generator=g, device=DEV, dtype=torch.float32).to(torch.bfloat16)


@pytest.mark.parametrize("S", [16384, 131072])
def test_attention_at_the_benchmark_sequence(ops, S):
    Hq, Hkv, D = 40, 8, 128
    G = Hq // Hkv
    q, k, v = randn_bf16((1, S, Hq, D), 50), randn_bf16((1, S, Hkv, D), 51), randn_bf16((1, S, Hkv, D), 52)
    out = ops.flash_attn(q, k, v, causal=True)
    torch.cuda.synchronize()
    rows = sample_rows(S, S, 72, seed=S)
    got = out[0, rows.to(DEV)].view(len(rows), Hkv, G, D)
    ref = torch.stack([oracle_rows(q[0, rows.to(DEV), g * G:(g + 1) * G], k[0], v[0], rows, g) for g in range(Hkv)], 1)
    check(f"attn_fwd_S{S}", got, ref, rows=len(rows), heads=Hq, note="single chunk, causal, 40:8, d=128; sampled rows vs fp32 oracle")


def test_attention_backward_at_16k(ops):
    """vita_flash_attn_bwd (dQ kernel + dK/dV kernel, reworked in round 2) at the 16K training sequence, 40:8 heads: every row of
    dQ / dK / dV of the first and the last kv group against fp32 autograd through oracle.attention.core_attention (evaluated as torch
    ops on the GPU: 5 heads x 16384^2 scores in fp32)."""
    S, Hq, Hkv, D = 16384, 40, 8, 128
    G = Hq // Hkv
    q, k, v = randn_bf16((1, S, Hq, D), 70), randn_bf16((1, S, Hkv, D), 71), randn_bf16((1, S, Hkv, D), 72)
    d_o = randn_bf16((1, S, Hq, D), 73)
    o, lse = ops.flash_attn(q, k, v, causal=True, return_lse=True)
    dq, dk, dv = ops.flash_attn_bwd(q, k, v, o, d_o, lse)
    torch.cuda.synchronize()
    got = {"dq": [], "dk": [], "dv": []}
    ref = {"dq": [], "dk": [], "dv": []}
    for g in (0, Hkv - 1):
        qf = q[0, :, g * G:(g + 1) * G].float().reshape(S, 1, G, D).requires_grad_(True)
        kf = k[0, :, g:g + 1].float().reshape(S, 1, 1, D).requires_grad_(True)
        vf = v[0, :, g:g + 1].float().reshape(S, 1, 1, D).requires_grad_(True)
        out = oattn.core_attention(qf, kf, vf, True)                                  # [S, 1, G * D] fp32
        out.backward(d_o[0, :, g * G:(g + 1) * G].float().reshape(S, 1, G * D))
        ref["dq"].append(qf.grad.reshape(S, G, D).cpu()); got["dq"].append(dq[0, :, g * G:(g + 1) * G].float().cpu())
        ref["dk"].append(kf.grad.reshape(S, D).cpu()); got["dk"].append(dk[0, :, g].float().cpu())
        ref["dv"].append(vf.grad.reshape(S, D).cpu()); got["dv"].append(dv[0, :, g].float().cpu())
        del qf, kf, vf, out
        torch.cuda.empty_cache()
    for name in ("dq", "dk", "dv"):
        check(f"attn_bwd_S{S}_{name}", torch.stack(got[name]), torch.stack(ref[name]), rows=S, groups=2,
              note="all rows of kv groups 0 and 7 vs fp32 autograd through the oracle attention")


@pytest.mark.parametrize("S,rank", [(131072, 3), (1048576, 5)])
def test_attention_cp8_chunk_tables_at_the_benchmark_geometry(ops, S, rank):
    """One rank, one kv-head split (2 kv heads, 10 query heads) of the CP = 8 layout: local queries in zig-zag order against
    the gathered buffer [rank p][K | V][S_l][2][128] through chunk tables (long_vita_amd/dot_product_attention.py:forward_cp)."""
    cp, hg, G, D = 8, 2, 5, 128
    C = S // (2 * cp)
    s_l = 2 * C
    k, v = randn_bf16((S, hg, D), 61), randn_bf16((S, hg, D), 62)                  # global order
    q = randn_bf16((1, s_l, hg * G, D), 60)                                        # the rank's local rows
    pos = glue.calibration_index(S, cp, rank)                                      # their global positions
    packed = torch.empty(cp, 2, s_l, hg, D, dtype=torch.bfloat16, device=DEV)
    for p in range(cp):
        idx = glue.calibration_index(S, cp, p).to(DEV)
        packed[p, 0], packed[p, 1] = k[idx], v[idx]
    rows_buf = packed.view(cp * 2 * s_l, hg, D)
    kv_gid, kv_row = [], []
    for p in range(cp):
        kv_gid += [p, 2 * cp - 1 - p]
        kv_row += [p * 2 * s_l, p * 2 * s_l + C]
    out = ops.flash_attn(q, rows_buf.unsqueeze(0), rows_buf[s_l:].unsqueeze(0), causal=True, chunk_len=C,
                         q_chunk_gid=[rank, 2 * cp - 1 - rank], kv_chunk_gid=kv_gid, kv_chunk_row=kv_row)
    torch.cuda.synchronize()
    rows = sample_rows(s_l, C, 64, seed=S + rank)
    got = out[0, rows.to(DEV)].view(len(rows), hg, G, D)
    ref = torch.stack([oracle_rows(q[0, rows.to(DEV), g * G:(g + 1) * G], k, v, pos[rows], g) for g in range(hg)], 1)
    check(f"attn_cp8_Sl{s_l}_S{S}", got, ref, rows=len(rows), rank=rank,
          note="zig-zag chunk tables, one kv-head split of CP = 8; sampled local rows vs fp32 oracle at their global positions")


def test_prefill_48_layers_full_width_16k(ops):
    """BASELINE config 2's decoder: 48 layers, hidden 5120, 40:8 heads, FFN 13824, vocabulary 152064, S = 16384.
    The host cannot evaluate the oracle at this size (0.56 PFLOP), so the SAME oracle functions (oracle.llm.decoder_layer,
    oracle.glue.*, oracle.attention.core_attention) run as torch ops on the GPU here — they stay the checker; the product
    path never touches them.  Two evaluations of the oracle:
      * `chain`: bf16 activations, i.e. the reference's own rounding chain (what Megatron computes in bf16);
      * `exact`: fp32 activations over the same bf16 weights — the function both bf16 evaluations approximate.
    Over 48 layers two correct bf16 evaluations drift apart by as much as each drifts from `exact` (rounding differences are
    amplified layer by layer), so the pinned statement is: the HIP logits are no further from `exact` than the reference's
    chain is (x 1.25), and the three distances are recorded."""
    from long_vita_amd import gpt_vl_model, synthetic
    S = 16384
    cfg = gpt_vl_model.GPTConfig()
    model = gpt_vl_model.GPTVLModel.random_init(cfg, seed=1234, device=DEV)
    tokens, _ = synthetic.make_request(S, 0, seed=7, device=DEV)
    g = torch.Generator().manual_seed(3)
    sel = sorted({0, 1, 255, 256, 8191, 8192, S - 2, S - 1} | {int(x) for x in torch.randint(0, S, (56,), generator=g)})
    mask = torch.zeros(1, S, dtype=torch.bool, device=DEV)
    mask[0, sel] = True
    logits = model.forward(tokens, logit_mask=mask)[0].float()                       # [n_sel, V]
    torch.cuda.synchronize()

    ocfg = ollm.LLMConfig()
    p = model.p                                                                      # same bf16 tensors, Megatron layout
    freqs = glue.rope_emb(S, glue.rope_inv_freq(ocfg.head_dim, ocfg.rope_theta)).to(DEV)

    def attn_fn(q, k, v):
        outs = []
        for grp in range(ocfg.kv_groups):                                           # one kv group at a time: 5 x S x S fp32 scores
            o = oattn.core_attention(q[:, :, grp * ocfg.qpg:(grp + 1) * ocfg.qpg], k[:, :, grp:grp + 1], v[:, :, grp:grp + 1], True)
            outs.append(o.view(S, 1, ocfg.qpg, ocfg.head_dim))
        return torch.cat(outs, 2).reshape(S, 1, -1)

    def oracle_logits(dtype):
        with torch.no_grad():
            h = glue.embedding_scatter(p["embed"][tokens].to(dtype), None)
            for lp in p["layers"]:
                h, _ = ollm.decoder_layer(h, lp, ocfg, freqs, attn_fn)
            h = glue.rmsnorm(h, p["final_ln"].to(dtype), ocfg.eps)
            return glue.masked_linear_fwd(h.float(), p["lm_head"].float(), None, mask)[:, 0]

    exact = oracle_logits(torch.float32)
    chain = oracle_logits(torch.bfloat16)
    rel = lambda a, b: float((a - b).norm() / b.norm())                              # noqa: E731
    e_hip, e_chain, e_pair = rel(logits, exact), rel(chain, exact), rel(logits, chain)
    record("prefill_48L_S16384_logits", hip_vs_exact_rel_l2=e_hip, reference_chain_vs_exact_rel_l2=e_chain,
           hip_vs_reference_chain_rel_l2=e_pair, hip_vs_exact_max_abs=float((logits - exact).abs().max()),
           exact_rms=float(exact.pow(2).mean().sqrt()), rows=len(sel), layers=cfg.num_layers,
           top1_agreement_hip=float((logits.argmax(-1) == exact.argmax(-1)).float().mean()),
           top1_agreement_chain=float((chain.argmax(-1) == exact.argmax(-1)).float().mean()),
           note="48-layer full-width text prefill, 64 sampled logit rows; exact = fp32 activations over the same bf16 weights, "
                "chain = the reference's bf16 rounding chain (oracle functions as torch ops on the GPU)")
    assert e_hip < 1.25 * e_chain + 1e-3, (e_hip, e_chain)
    assert e_pair < 2.5 * e_chain + 1e-3, (e_pair, e_chain)
