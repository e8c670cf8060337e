"""Parity at the BENCHMARK configurations (VERDICT r1 "next round" item 1), with the achieved errors recorded.

  * vita_flash_attn_fwd at S = 16384 and 131072 (40:8 heads, d = 128, causal) — BASELINE configs 2 / 3 on one GPU;
  * the CP = 8 chunk-table geometries of configs 3 and 4 for one rank and one kv-head split: S_l = 16384 local queries
    against 131072 gathered keys, S_l = 131072 against 1 M keys (the packed [rank][K | V][S_l][hg][d] buffer that
    dot_product_attention.forward_cp hands to the kernel);
  * one full-width (5120 / 40:8 / 13824) 48-layer prefill at 16K.
The oracle is evaluated on SAMPLED query rows only (first / last row of chunks and 256-row tiles, 64-key tile edges,
random rows): fp32 `oracle.attention.core_attention` on the CPU with the rows' global positions — restating
M/core/transformer/dot_product_attention.py:186-289 with the zig-zag ownership of M/training/utils.py:329-341.
Every achieved rel-L2 / max-abs lands in gpurun_out/r06_parity.json (copied to profiles/; r03_parity.json is the previous round's record); each limit below is
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
    # r03: against the reference's own bf16 dtype chain (Megatron's unfused attention run in bf16, oracle chain=True) and, for scale,
    # that chain against fp32 math.  Measured: forward HIP-vs-chain 2.08e-3 / 2.18e-3 (16K / 128K) where chain-vs-exact is 1.77e-3 /
    # 1.83e-3 and HIP-vs-exact 1.19e-3 — the reference's chain is FURTHER from fp32 math than this kernel is; backward HIP-vs-chain
    # 5.2e-3 / 5.5e-3 / 4.8e-3 (dq / dk / dv), chain-vs-exact 4.6e-3 / 5.0e-3 / 4.3e-3, HIP-vs-exact 2.4e-3.
    "attn_fwd_S16384_vs_chain": (3.3e-3, 2.4e-2), "attn_fwd_S131072_vs_chain": (3.3e-3, 2.4e-2),
    "attn_fwd_S16384_chain_vs_exact": (2.8e-3, 2.2e-2), "attn_fwd_S131072_chain_vs_exact": (2.8e-3, 2.2e-2),
    "attn_bwd_S16384_dq_vs_chain": (7.9e-3, 2.4e-2), "attn_bwd_S16384_dk_vs_chain": (8.3e-3, 4.7e-2),
    "attn_bwd_S16384_dv_vs_chain": (7.2e-3, 9.4e-2),
    "attn_bwd_S16384_dq_chain_vs_exact": (7.0e-3, 2.1e-2), "attn_bwd_S16384_dk_chain_vs_exact": (7.5e-3, 3.1e-2),
    "attn_bwd_S16384_dv_chain_vs_exact": (6.4e-3, 5.5e-2),
}


def record(name, **metrics):
    record_parity(name, **metrics)


PROVISIONAL = (2e-2, 1.0)        # a case measured for the first time: recorded, asserted loosely, then pinned in LIMITS


def check(name, got, ref, **extra):
    got, ref = got.float().cpu(), ref.float().cpu()
    rel = float((got - ref).norm() / ref.norm())
    mab = float((got - ref).abs().max())
    lim = LIMITS.get(name, PROVISIONAL)
    record(name, rel_l2=rel, max_abs=mab, ref_rms=float(ref.pow(2).mean().sqrt()), limit_rel_l2=lim[0],
           limit_max_abs=lim[1], pinned=name in LIMITS, **extra)
    assert rel < lim[0], (name, rel)
    assert mab < lim[1], (name, mab)


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


def oracle_rows(q_rows, k, v, q_pos, group, chain=False):
    """q_rows [n, G, d] (the G query heads of kv group `group`), k / v [S, ng, d] in global order -> [n, G, d], on the CPU.
    chain=False: fp32 math (`exact`).  chain=True: the reference's own bf16 dtype chain — Megatron's unfused attention run in bf16
    (bf16 scores, fp32 softmax cast to bf16, bf16 context; oracle.attention.core_attention(chain=True), pinned bit-for-bit against
    the reference's wrapper by tests/golden/unfused_attention_bf16.pt)."""
    n, G, d = q_rows.shape
    cast = (lambda t: t.cpu()) if chain else (lambda t: t.float().cpu())
    out = oattn.core_attention(cast(q_rows).view(n, 1, G, d), cast(k[:, group:group + 1]).unsqueeze(1),
                               cast(v[:, group:group + 1]).unsqueeze(1), True, q_pos=q_pos.cpu(), chain=chain)
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
    chain = torch.stack([oracle_rows(q[0, rows.to(DEV), g * G:(g + 1) * G], k[0], v[0], rows, g, chain=True) for g in range(Hkv)], 1)
    check(f"attn_fwd_S{S}_vs_chain", got, chain, rows=len(rows), heads=Hq,
          note="the same rows vs the reference's bf16 dtype chain (Megatron's unfused attention in bf16)")
    check(f"attn_fwd_S{S}_chain_vs_exact", chain, ref, note="how far the reference's own bf16 chain is from fp32 math on these rows")


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
    chain = {"dq": [], "dk": [], "dv": []}
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
        # the reference's bf16 chain: bf16 leaves, bf16 scores / probabilities, gradients rounded where bf16 autograd rounds them
        qb = q[0, :, g * G:(g + 1) * G].reshape(S, 1, G, D).clone().requires_grad_(True)
        kb = k[0, :, g:g + 1].reshape(S, 1, 1, D).clone().requires_grad_(True)
        vb = v[0, :, g:g + 1].reshape(S, 1, 1, D).clone().requires_grad_(True)
        oattn.core_attention(qb, kb, vb, True, chain=True).backward(d_o[0, :, g * G:(g + 1) * G].reshape(S, 1, G * D))
        chain["dq"].append(qb.grad.reshape(S, G, D).float().cpu()); chain["dk"].append(kb.grad.reshape(S, D).float().cpu())
        chain["dv"].append(vb.grad.reshape(S, D).float().cpu())
        del qb, kb, vb
        torch.cuda.empty_cache()
    for name in ("dq", "dk", "dv"):
        check(f"attn_bwd_S{S}_{name}", torch.stack(got[name]), torch.stack(ref[name]), rows=S, groups=2,
              note="all rows of kv groups 0 and 7 vs fp32 autograd through the oracle attention")
        check(f"attn_bwd_S{S}_{name}_vs_chain", torch.stack(got[name]), torch.stack(chain[name]), rows=S, groups=2,
              note="the same vs bf16 autograd through the reference's bf16 dtype chain")
        check(f"attn_bwd_S{S}_{name}_chain_vs_exact", torch.stack(chain[name]), torch.stack(ref[name]),
              note="the reference's own bf16 chain vs fp32 math")


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


@pytest.mark.parametrize("std", [0.02, 0.01])
def test_prefill_48_layers_full_width_16k(ops, std):
    """BASELINE config 2's decoder: 48 layers, hidden 5120, 40:8 heads, FFN 13824, vocabulary 152064, S = 16384.
    The host cannot evaluate the oracle at this size (0.56 PFLOP), so the SAME oracle functions (oracle.llm.decoder_layer,
    oracle.glue.*, oracle.attention.core_attention) run as torch ops on the GPU here — they stay the checker; the product
    path never touches them.  Two evaluations of the oracle:
      * `chain`: bf16 activations, i.e. the reference's own rounding chain (what Megatron computes in bf16);
      * `exact`: fp32 activations over the same bf16 weights — the function both bf16 evaluations approximate.
    Over 48 layers two correct bf16 evaluations drift apart by as much as each drifts from `exact` (rounding differences are
    amplified layer by layer), so the pinned statement is: the HIP logits are no further from `exact` than the reference's
    chain is (x 1.25), and the three distances are recorded.
    std: the weight scale.  0.02 is SURVEY.md 8d's synthetic choice; 0.01 is the reference's own `--init-method-std` (stage3 .sh:176),
    at which 48 random layers amplify rounding differences far less — the scale at which "HIP vs chain" is a meaningful number."""
    from long_vita_amd import gpt_vl_model, synthetic
    S = 16384
    cfg = gpt_vl_model.GPTConfig()
    model = gpt_vl_model.GPTVLModel.random_init(cfg, seed=1234, device=DEV, std=std)
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
            o = oattn.core_attention(q[:, :, grp * ocfg.qpg:(grp + 1) * ocfg.qpg], k[:, :, grp:grp + 1], v[:, :, grp:grp + 1], True,
                                     chain=True)            # bf16 activations: Megatron's unfused dtype chain; fp32: plain fp32 math
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
    record("prefill_48L_S16384_logits" + ("" if std == 0.02 else f"_std{std}"), weight_std=std, hip_vs_exact_rel_l2=e_hip, reference_chain_vs_exact_rel_l2=e_chain,
           hip_vs_reference_chain_rel_l2=e_pair, hip_vs_exact_max_abs=float((logits - exact).abs().max()),
           exact_rms=float(exact.pow(2).mean().sqrt()), rows=len(sel), layers=cfg.num_layers,
           top1_agreement_hip=float((logits.argmax(-1) == exact.argmax(-1)).float().mean()),
           top1_agreement_chain=float((chain.argmax(-1) == exact.argmax(-1)).float().mean()),
           note="48-layer full-width text prefill, 64 sampled logit rows; exact = fp32 activations over the same bf16 weights, "
                "chain = the reference's bf16 rounding chain (oracle functions as torch ops on the GPU)")
    assert e_hip < 1.25 * e_chain + 1e-3, (e_hip, e_chain)
    # measured (profiles/r03_parity.json): std 0.02: HIP 9.7e-2, chain 1.32e-1 from exact, 1.18e-1 from each other (top-1 agreement with
    # exact: HIP 83 %, chain 73 %); std 0.01 (the reference's init): HIP 1.93e-2, chain 1.93e-2, pair 2.2e-2 (top-1 95 % / 95 %)
    assert e_pair < 1.5 * e_chain + 1e-3, (e_pair, e_chain)


def _group_attn(ocfg, S, chain):
    def attn_fn(q, k, v):
        outs = []
        for grp in range(ocfg.kv_groups):                                               # one kv group at a time: 5 x S x S scores
            o = oattn.core_attention(q[:, :, grp * ocfg.qpg:(grp + 1) * ocfg.qpg], k[:, :, grp:grp + 1], v[:, :, grp:grp + 1], True, chain=chain)
            outs.append(o.view(S, 1, ocfg.qpg, ocfg.head_dim))
        return torch.cat(outs, 2).reshape(S, 1, -1)
    return attn_fn


def test_full_width_decoder_layer_forward_and_every_gradient_at_16k(ops):
    """One decoder layer at the 14B width (5120 / 40:8 / 13824) and S = 16384: the fused forward (GPTVLModel.decoder_layer) and the
    whole backward sweep of training.TrainStep._layer_backward (recompute + vita_flash_attn_bwd + dgrad / wgrad GEMMs + RMSNorm /
    SwiGLU / RoPE backward) — output, input gradient and all seven parameter gradients — against torch autograd over the oracle layer
    (oracle.llm.decoder_layer as torch ops on the GPU) evaluated twice: `exact` (fp32 activations over the same bf16 weights) and
    `chain` (bf16 activations, the reference's rounding chain incl. Megatron's unfused bf16 attention).  VERDICT r2: "full-width
    training gradients are pinned nowhere"."""
    from long_vita_amd import gpt_vl_model, training
    S = 16384
    cfg = gpt_vl_model.GPTConfig(num_layers=1, vocab=1024)
    model = gpt_vl_model.GPTVLModel.random_init(cfg, seed=77, device=DEV)
    lp = model.p["layers"][0]
    gen = torch.Generator(device=DEV).manual_seed(78)
    lp["ln1"].copy_((1 + 0.1 * torch.randn(cfg.hidden, generator=gen, device=DEV)).bfloat16())
    lp["ln2"].copy_((1 + 0.1 * torch.randn(cfg.hidden, generator=gen, device=DEV)).bfloat16())
    h = torch.randn(S, cfg.hidden, generator=gen, device=DEV).bfloat16()
    dh = (torch.randn(S, cfg.hidden, generator=gen, device=DEV) * 0.1).bfloat16()
    cos, sin = model.rotary_pos_emb(S)
    out = model.decoder_layer(h.clone(), lp, cos, sin, model._workspace(S, h.device)).clone()
    g = {}
    with torch.no_grad():
        dx = training.TrainStep(model)._layer_backward(dh, h, lp, cos, sin, g)
    torch.cuda.synchronize()
    got = dict(g, out=out, dx=dx)
    model._ws = {}
    ocfg = ollm.LLMConfig(num_layers=1, vocab=1024)
    freqs = glue.rope_emb(S, glue.rope_inv_freq(ocfg.head_dim, ocfg.rope_theta)).to(DEV)
    refs = {}
    for mode, dt in (("exact", torch.float32), ("chain", torch.bfloat16)):
        x = h.view(S, 1, -1).to(dt).clone().requires_grad_(True)
        lpo = {k: v.detach().to(dt).clone().requires_grad_(True) for k, v in lp.items()}
        y, _ = ollm.decoder_layer(x, lpo, ocfg, freqs, _group_attn(ocfg, S, chain=dt != torch.float32))
        y.backward(dh.view(S, 1, -1).to(dt))
        refs[mode] = dict({k: v.grad.float() for k, v in lpo.items()}, out=y.detach().float().view(S, -1), dx=x.grad.float().view(S, -1))
        del x, lpo, y
        torch.cuda.empty_cache()
    rel = lambda a, b: float((a.float() - b.float()).norm() / b.float().norm())        # noqa: E731
    rec = {}
    for k in ("out", "dx", "qkv_w", "qkv_b", "o_w", "fc1_w", "fc2_w", "ln1", "ln2"):
        rec[k] = dict(hip_vs_exact=rel(got[k], refs["exact"][k]), hip_vs_chain=rel(got[k], refs["chain"][k]),
                      chain_vs_exact=rel(refs["chain"][k], refs["exact"][k]))
    record("decoder_layer_full_width_S16384", **rec, note="rel-L2 per tensor; exact = fp32 activations, chain = the reference's bf16 chain")
    for k, r in rec.items():
        # the HIP result is no further from fp32 math than the reference's own bf16 chain is (x 1.5), tensor by tensor
        assert r["hip_vs_exact"] < 1.5 * r["chain_vs_exact"] + 1e-3, (k, r)
        assert r["hip_vs_chain"] < LAYER_LIMITS.get(k, 3e-2), (k, r)


# HIP vs the reference's bf16 chain, per tensor, 1.5 x the values measured on the MI355X (profiles/r03_parity.json): out 6.8e-3,
# dx 8.8e-3, qkv_w 1.30e-2, qkv_b 7.6e-3, o_w 1.27e-2, fc1_w 6.5e-3, fc2_w 6.7e-3, ln1 1.31e-2, ln2 6.9e-3 (the chain itself is
# 7.5e-3 .. 1.8e-2 from fp32 math on the same tensors, the HIP path 6.8e-3 .. 1.5e-2)
LAYER_LIMITS = {"out": 1.02e-2, "dx": 1.32e-2, "qkv_w": 1.95e-2, "qkv_b": 1.15e-2, "o_w": 1.91e-2, "fc1_w": 9.8e-3, "fc2_w": 1.01e-2,
                "ln1": 1.97e-2, "ln2": 1.03e-2}


def test_vit_scatter_layer0_at_128k_with_506_frames(ops):
    """The HEADLINE configuration checked against the oracle, not against itself: 506 synthetic frames through the full
    24-layer InternViT-300M + projector, scattered into the 131072-token request (long_vita_amd.synthetic: BASELINE config 3's
    layout), then decoder layer 0 at full width — on SAMPLED rows (first / last rows, frame boundaries, tile edges, text tail):
    the ViT features of sampled frames, the embedded rows, and the layer-0 output rows, each against `exact` (fp32 activations) and
    `chain` (bf16 activations) evaluations of the oracle's functions run as torch ops on the GPU.  Layer 0 of the oracle needs the
    K / V of all 131072 rows (hence the whole ViT), but Q, the o-projection and the MLP only of the sampled rows."""
    from long_vita_amd import gpt_vl_model, synthetic, vision
    from oracle import vit as ovit
    S, n_frames = 131072, 506
    vcfg = ovit.ViTConfig()
    vp = ovit.init_vit_params(vcfg, seed=1234)
    vis = vision.MegatronVisionModel.from_oracle_layout(vision.VisionConfig(), vp, DEV)
    cfg = gpt_vl_model.GPTConfig(num_layers=1)
    model = gpt_vl_model.GPTVLModel.random_init(cfg, seed=1234, device=DEV, external_feature_model=vis)
    tokens, ext = synthetic.make_request(S, n_frames, seed=1234, device=DEV)
    with torch.no_grad():
        feats = vis(images=ext["images"])                                                  # [506, 256, 5120]
        h0 = model.embedding(tokens, None, external_feature_dict={"features": feats, "indices": ext["indices"]}).view(S, cfg.hidden)
        cos, sin = model.rotary_pos_emb(S)
        h1 = model.decoder_layer(h0.clone(), model.p["layers"][0], cos, sin, model._workspace(S, h0.device)).clone()
    torch.cuda.synchronize()
    model._ws = {}
    last_vis = n_frames * 258
    rows = sorted({0, 1, 2, 256, 257, 258, 259, 514, 65535, 65536, 65791, 65792, last_vis - 2, last_vis - 1, last_vis, last_vis + 1,
                   S - 65, S - 64, S - 2, S - 1} | set(sample_rows(S, S, 44, seed=9).tolist()))
    rows_t = torch.tensor(rows, device=DEV)
    ocfg = ollm.LLMConfig(num_layers=1)
    lp = model.p["layers"][0]
    vpd = {k: (v.to(DEV) if torch.is_tensor(v) else [{kk: vv.to(DEV) for kk, vv in l_.items()} for l_ in v]) for k, v in vp.items()}
    freqs = glue.rope_emb(S, glue.rope_inv_freq(ocfg.head_dim, ocfg.rope_theta)).to(DEV)
    out = {}
    for mode, dt in (("exact", torch.float32), ("chain", torch.bfloat16)):
        with torch.no_grad():
            f = torch.cat([ovit.vision_model(ch.to(dt), vpd, vcfg) for ch in torch.split(ext["images"], 23, dim=0)], 0)   # [506, 256, 5120]
            we = model.p["embed"][tokens].to(dt)
            e0 = glue.embedding_scatter(we, {"features": f, "indices": ext["indices"]})        # [S, 1, h]
            x = glue.rmsnorm(e0, lp["ln1"].to(dt), ocfg.eps)
            q, k, v = ollm.split_qkv(ovit.linear(x, lp["qkv_w"], lp["qkv_b"]), ocfg)
            q = glue.apply_rotary_pos_emb_bshd(q[rows_t], freqs[rows_t])
            k = glue.apply_rotary_pos_emb_bshd(k, freqs)
            ctx = torch.cat([oattn.core_attention(q[:, :, gi * ocfg.qpg:(gi + 1) * ocfg.qpg], k[:, :, gi:gi + 1], v[:, :, gi:gi + 1], True,
                                                  q_pos=rows_t, chain=dt != torch.float32).view(len(rows), 1, ocfg.qpg, ocfg.head_dim)
                             for gi in range(ocfg.kv_groups)], 2).reshape(len(rows), 1, -1)
            hm = e0[rows_t] + ovit.linear(ctx, lp["o_w"])
            y = ovit.linear(glue.rmsnorm(hm, lp["ln2"].to(dt), ocfg.eps), lp["fc1_w"])
            gate, up = torch.chunk(y, 2, dim=-1)
            o1 = hm + ovit.linear(torch.nn.functional.silu(gate.float()).to(y.dtype) * up, lp["fc2_w"])
            out[mode] = dict(feats=f[::23].float(), embed=e0[rows_t, 0].float(), layer0=o1[:, 0].float())
        del f, we, e0, x, q, k, v, ctx, hm, y, o1
        torch.cuda.empty_cache()
    got = dict(feats=feats[::23].float(), embed=h0[rows_t].float(), layer0=h1[rows_t].float())
    rel = lambda a, b: float((a - b).norm() / b.norm())                                        # noqa: E731
    rec = {}
    for k_ in ("feats", "embed", "layer0"):
        rec[k_] = dict(hip_vs_exact=rel(got[k_], out["exact"][k_]), hip_vs_chain=rel(got[k_], out["chain"][k_]),
                       chain_vs_exact=rel(out["chain"][k_], out["exact"][k_]),
                       hip_vs_chain_max_abs=float((got[k_] - out["chain"][k_]).abs().max()), ref_rms=float(out["exact"][k_].pow(2).mean().sqrt()))
    record("vit_scatter_layer0_S131072_506frames", **rec, rows=len(rows), frames_sampled=len(range(0, n_frames, 23)),
           note="feats = ViT + projector output of every 23rd frame; embed = embedded rows after the visual scatter; layer0 = decoder "
                "layer 0 output rows; rel-L2 vs exact (fp32 activations) / chain (bf16 activations) oracle evaluations on the GPU")
    for k_, r in rec.items():
        assert r["hip_vs_exact"] < 1.5 * r["chain_vs_exact"] + 2e-3, (k_, r)
        assert r["hip_vs_chain"] < HEADLINE_LIMITS.get(k_, 5e-2), (k_, r)


# measured: feats 8.30e-3, embed 8.29e-3, layer0 1.05e-2 (HIP vs chain); chain vs exact 1.28e-2 / 1.28e-2 / 1.30e-2, HIP vs exact
# 1.29e-2 / 1.27e-2 / 1.26e-2: 24 ViT layers + projector + one decoder layer in bf16 sit 1.3e-2 from fp32 math whoever computes them
HEADLINE_LIMITS = {"feats": 1.25e-2, "embed": 1.25e-2, "layer0": 1.6e-2}
