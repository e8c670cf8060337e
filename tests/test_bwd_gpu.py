"""Backward kernels vs torch autograd over the CPU oracle (the reference obtains these gradients from
autograd over the same forward expressions)."""
import math

import pytest
import torch

pytestmark = pytest.mark.gpu

from oracle import attention as oattn  # noqa: E402
from oracle import glue  # noqa: E402

from conftest import tol  # noqa: E402

DEV = "cuda"


@pytest.fixture(scope="module")
def ops():
    from long_vita_amd import ops as _ops
    _ops._L.load(allow_build=False)
    return _ops


def g(seed):
    return torch.Generator().manual_seed(seed)


def rel_l2(a, b):
    a, b = a.float().cpu(), b.float().cpu()
    return float((a - b).norm() / (b.norm() + 1e-30))


@pytest.mark.parametrize("R,C", [(64, 64), (100, 130), (5120, 7168), (1, 77), (333, 8)])
def test_transpose_bit_exact(ops, R, C):
    x = torch.randn(R, C, generator=g(1)).bfloat16()
    assert torch.equal(ops.transpose(x.to(DEV)).cpu(), x.t().contiguous())
    big = torch.randn(R, C + 16, generator=g(2)).bfloat16().to(DEV)
    assert torch.equal(ops.transpose(big[:, 8:8 + C]).cpu(), big[:, 8:8 + C].t().contiguous().cpu())


@pytest.mark.parametrize("rows,cols", [(37, 5120), (300, 1024)])
def test_rmsnorm_bwd(ops, rows, cols):
    x = (torch.randn(rows, cols, generator=g(3)) * 2).bfloat16().requires_grad_(True)
    w = (1 + 0.1 * torch.randn(cols, generator=g(4))).bfloat16().requires_grad_(True)
    dy = torch.randn(rows, cols, generator=g(5)).bfloat16()
    glue.rmsnorm(x, w, 1e-6).backward(dy)
    dw = torch.zeros(cols, dtype=torch.float32, device=DEV)
    dx = ops.rmsnorm_bwd(dy.to(DEV), x.detach().to(DEV), w.detach().to(DEV), 1e-6, dw)
    tol("dx, x.grad", rel_l2(dx, x.grad), 4e-3)
    tol("dw, w.grad", rel_l2(dw, w.grad), 2.6e-03)                     # autograd sums bf16 products in bf16 storage


def test_swiglu_fwd_bwd(ops):
    rows, F = 70, 13824
    y = torch.randn(rows, 2 * F, generator=g(6)).bfloat16().requires_grad_(True)
    da = torch.randn(rows, F, generator=g(7)).bfloat16()
    gate, up = torch.chunk(y, 2, dim=-1)
    a = torch.nn.functional.silu(gate.float()).to(y.dtype) * up
    a.backward(da)
    out = ops.swiglu(y.detach().to(DEV))
    tol("out, a.detach()", rel_l2(out, a.detach()), 3e-3)
    dy = ops.swiglu_bwd(y.detach().to(DEV), da.to(DEV))
    tol("dy, y.grad", rel_l2(dy, y.grad), 4e-3)


def test_gelu_bwd_and_layernorm_param_grad(ops):
    x = torch.randn(64, 1024, generator=g(8)).bfloat16().requires_grad_(True)
    dy = torch.randn(64, 1024, generator=g(9)).bfloat16()
    torch.nn.functional.gelu(x.float()).to(x.dtype).backward(dy)
    tol("ops.gelu_bwd(x.detach().to(DEV), dy.to(DEV)), x.grad", rel_l2(ops.gelu_bwd(x.detach().to(DEV), dy.to(DEV)), x.grad), 4e-3)
    rows, cols = 200, 4096
    xi = (torch.randn(rows, cols, generator=g(10)) + 0.3).bfloat16()
    gam = torch.ones(cols, requires_grad=True)
    bet = torch.zeros(cols, requires_grad=True)
    d2 = torch.randn(rows, cols, generator=g(11)).bfloat16()
    torch.nn.functional.layer_norm(xi.float(), (cols,), gam, bet, 1e-5).backward(d2.float())
    dg = torch.zeros(cols, dtype=torch.float32, device=DEV)
    db = torch.zeros(cols, dtype=torch.float32, device=DEV)
    ops.layernorm_param_grad(d2.to(DEV), xi.to(DEV), dg, db, 1e-5)
    torch.testing.assert_close(dg.cpu(), gam.grad, rtol=2e-3, atol=2e-3)
    torch.testing.assert_close(db.cpu(), bet.grad, rtol=2e-3, atol=2e-3)


def test_ce_loss_and_grad(ops):
    n, V = 9, 152064
    logits = (torch.randn(n, V, generator=g(12)) * 3).bfloat16()
    labels = torch.randint(0, V, (n,), generator=g(13))
    lf = logits.float().requires_grad_(True)
    ref = torch.nn.functional.cross_entropy(lf, labels, reduction="none")
    scale = torch.rand(n, generator=g(14)) + 0.5
    (ref * scale).sum().backward()
    loss, dl = ops.ce_loss(logits.to(DEV), labels.to(DEV), scale.to(DEV), want_grad=True)
    torch.testing.assert_close(loss.cpu(), ref.detach(), rtol=1e-5, atol=1e-5)
    tol("dl, lf.grad", rel_l2(dl, lf.grad), 1.9e-03)
    with pytest.raises(IndexError):                               # strict (the stand-alone step: labels are always real tokens)
        ops.ce_loss(logits.to(DEV), torch.full((n,), V, dtype=torch.int64, device=DEV))
    # strict=False: Megatron's masked target for the -100 padding — loss = log sum exp(l - max), gradient = softmax * scale, no raise
    lab2 = labels.clone()
    lab2[[1, 5]] = -100
    loss2, dl2 = ops.ce_loss(logits.to(DEV), lab2.to(DEV), scale.to(DEV), want_grad=True, strict=False)
    lf2 = logits.float()
    want = (lf2 - lf2.max(-1, keepdim=True)[0]).exp().sum(-1).log()
    torch.testing.assert_close(loss2.cpu()[[1, 5]], want[[1, 5]], rtol=1e-5, atol=1e-5)
    torch.testing.assert_close(loss2.cpu()[[0, 2, 3, 4, 6, 7, 8]], ref.detach()[[0, 2, 3, 4, 6, 7, 8]], rtol=1e-5, atol=1e-5)
    sm = torch.softmax(lf2[[1, 5]], -1) * scale[[1, 5], None]
    tol("masked rows: softmax * scale", rel_l2(dl2[[1, 5]], sm), 1.9e-03)


def test_row_scatter_add(ops):
    src = torch.randn(500, 5120, generator=g(15)).bfloat16()
    idx = torch.randint(0, 64, (500,), generator=g(16))
    idx[::7] = -1                                                   # skipped rows
    dst = torch.zeros(64, 5120, dtype=torch.float32, device=DEV)
    ops.row_scatter_add_f32_(dst, idx.to(DEV), src.to(DEV))
    keep = idx >= 0
    ref = torch.zeros(64, 5120).index_add_(0, idx[keep], src[keep].float())
    torch.testing.assert_close(dst.cpu(), ref, rtol=1e-5, atol=1e-4)


# ---------------------------------------------------------------------------------------------
def _attn_grads_ref(q, k, v, d_o, q_pos=None, k_pos=None):
    """autograd through the unfused reference math (fp32) on bf16-valued inputs."""
    qf = q.float().requires_grad_(True)
    kf = k.float().requires_grad_(True)
    vf = v.float().requires_grad_(True)
    o = oattn.core_attention(qf.transpose(0, 1), kf.transpose(0, 1), vf.transpose(0, 1), True, q_pos=q_pos, k_pos=k_pos)
    B, S, H, Dh = q.shape
    o = o.view(S, B, H, Dh).transpose(0, 1)
    o.backward(d_o.float())
    return o.detach(), qf.grad, kf.grad, vf.grad


@pytest.mark.parametrize("S,Hq,Hkv,Dh", [(128, 2, 1, 128), (256, 5, 1, 128), (512, 10, 2, 128), (1024, 5, 1, 128), (384, 4, 2, 96), (1024, 3, 3, 96),
                                        (256, 4, 2, 64)])
def test_flash_attn_bwd_single_chunk(ops, S, Hq, Hkv, Dh):
    q = torch.randn(1, S, Hq, Dh, generator=g(20)).bfloat16()
    k = torch.randn(1, S, Hkv, Dh, generator=g(21)).bfloat16()
    v = torch.randn(1, S, Hkv, Dh, generator=g(22)).bfloat16()
    d_o = torch.randn(1, S, Hq, Dh, generator=g(23)).bfloat16()
    _, dq_r, dk_r, dv_r = _attn_grads_ref(q, k, v, d_o)
    qd, kd, vd = q.to(DEV), k.to(DEV), v.to(DEV)
    o, lse = ops.flash_attn(qd, kd, vd, causal=True, return_lse=True)
    dq, dk, dv = ops.flash_attn_bwd(qd, kd, vd, o, d_o.to(DEV), lse)
    # P and dS are rounded to bf16 before their MFMAs (as flash-attn does): 1.5e-2 relative L2
    tol("dv, dv_r", rel_l2(dv, dv_r), 3.5e-03)
    tol("dq, dq_r", rel_l2(dq, dq_r), 4.1e-03)
    tol("dk, dk_r", rel_l2(dk, dk_r), 4.0e-03)


def test_flash_attn_bwd_mixed_qkv_layout(ops):
    """Gradients written straight into a mixed-QKV-shaped buffer (grouped dq view, dk/dv head slots)."""
    S, ng, qpg, d = 256, 2, 5, 128
    mixed = torch.randn(1, S, ng, qpg + 2, d, generator=g(24)).bfloat16().to(DEV)
    q5, kview, vview = mixed[:, :, :, :qpg], mixed[:, :, :, qpg], mixed[:, :, :, qpg + 1]
    d_o = torch.randn(1, S, ng * qpg, d, generator=g(25)).bfloat16().to(DEV)
    o, lse = ops.flash_attn(q5, kview, vview, causal=True, return_lse=True)
    dmixed = torch.zeros_like(mixed)
    ops.flash_attn_bwd(q5, kview, vview, o, d_o, lse, dq5=dmixed[:, :, :, :qpg], dk=dmixed[:, :, :, qpg],
                       dv=dmixed[:, :, :, qpg + 1])
    _, dq_r, dk_r, dv_r = _attn_grads_ref(q5.reshape(1, S, ng * qpg, d).cpu(), kview.cpu(), vview.cpu(), d_o.cpu())
    tol("dmixed[:, :, :, :qpg].reshape(1, S, ng * qpg, d), dq_r", rel_l2(dmixed[:, :, :, :qpg].reshape(1, S, ng * qpg, d), dq_r), 3.9e-03)
    tol("dmixed[:, :, :, qpg], dk_r", rel_l2(dmixed[:, :, :, qpg], dk_r), 3.8e-03)
    tol("dmixed[:, :, :, qpg + 1], dv_r", rel_l2(dmixed[:, :, :, qpg + 1], dv_r), 3.5e-03)


@pytest.mark.parametrize("cp,S", [(2, 1024), (4, 2048)])
def test_flash_attn_bwd_zigzag(ops, cp, S):
    """Per-rank backward over the gathered K/V (dk/dv in gathered layout), summed over ranks ==
    monolithic causal attention gradients (the reduce-scatter of the CP backward)."""
    Hq, Hkv, Dh = 5, 1, 128
    C = S // (2 * cp)
    q = torch.randn(1, S, Hq, Dh, generator=g(30)).bfloat16()
    k = torch.randn(1, S, Hkv, Dh, generator=g(31)).bfloat16()
    v = torch.randn(1, S, Hkv, Dh, generator=g(32)).bfloat16()
    d_o = torch.randn(1, S, Hq, Dh, generator=g(33)).bfloat16()
    _, dq_r, dk_r, dv_r = _attn_grads_ref(q, k, v, d_o)
    k_g = torch.cat([glue.zigzag_slice(k, cp, r) for r in range(cp)], 1).to(DEV)
    v_g = torch.cat([glue.zigzag_slice(v, cp, r) for r in range(cp)], 1).to(DEV)
    kv_gid, kv_row = [], []
    for r in range(cp):
        kv_gid += [r, 2 * cp - 1 - r]
        kv_row += [2 * r * C, (2 * r + 1) * C]
    dk_sum = torch.zeros(1, S, Hkv, Dh)
    dv_sum = torch.zeros(1, S, Hkv, Dh)
    for r in range(cp):
        geo = dict(chunk_len=C, q_chunk_gid=[r, 2 * cp - 1 - r], kv_chunk_gid=kv_gid, kv_chunk_row=kv_row)
        q_l = glue.zigzag_slice(q, cp, r).to(DEV)
        do_l = glue.zigzag_slice(d_o, cp, r).to(DEV)
        o, lse = ops.flash_attn(q_l, k_g, v_g, causal=True, return_lse=True, **geo)
        dq, dk, dv = ops.flash_attn_bwd(q_l, k_g, v_g, o, do_l, lse, **geo)
        tol("dq, glue.zigzag_slice(dq_r, cp, r)", rel_l2(dq, glue.zigzag_slice(dq_r, cp, r)), 3.8e-03)
        dk_sum += dk.float().cpu()
        dv_sum += dv.float().cpu()
    # un-zig-zag the gathered layout: buffer rows of rank p = zigzag_slice(., cp, p)
    dk_ref_g = torch.cat([glue.zigzag_slice(dk_r, cp, r) for r in range(cp)], 1)
    dv_ref_g = torch.cat([glue.zigzag_slice(dv_r, cp, r) for r in range(cp)], 1)
    tol("dk_sum, dk_ref_g", rel_l2(dk_sum, dk_ref_g), 3.8e-03)
    tol("dv_sum, dv_ref_g", rel_l2(dv_sum, dv_ref_g), 3.6e-03)


def test_cp8_dkv_reduce_scatter_in_bf16_ring_order_costs_what_a_bf16_sum_costs(ops):
    """VERDICT r2 (weak, parity): `ncclReduceScatter` of dK / dV in bf16 over 8 ranks adds a rounding the CP = 1 path does not
    have (cp_attn.hip vita_cp_attn_bwd; autograd_fns.FlashAttnCPFn.backward); the simulated-rank tests sum in fp32.  Here the eight
    ranks' partial dK / dV (gathered layout, bf16 as the kernels write them) are reduced the way a ring reduce-scatter does it:
    the segment of rank p starts at rank p + 1 and every hop adds the next rank's bf16 partial and rounds to bf16 (seven
    roundings).  Recorded and bounded: the ring result, the fp32-sum-then-round result, and the CP = 1 kernels, all against fp32
    autograd over the unsharded attention."""
    from conftest import record_parity
    cp, S, Hq, Hkv, Dh = 8, 4096, 10, 2, 128
    C = S // (2 * cp)
    q = torch.randn(1, S, Hq, Dh, generator=g(60)).bfloat16()
    k = torch.randn(1, S, Hkv, Dh, generator=g(61)).bfloat16()
    v = torch.randn(1, S, Hkv, Dh, generator=g(62)).bfloat16()
    d_o = torch.randn(1, S, Hq, Dh, generator=g(63)).bfloat16()
    _, _, dk_r, dv_r = _attn_grads_ref(q, k, v, d_o)
    k_g = torch.cat([glue.zigzag_slice(k, cp, r) for r in range(cp)], 1).to(DEV)
    v_g = torch.cat([glue.zigzag_slice(v, cp, r) for r in range(cp)], 1).to(DEV)
    kv_gid, kv_row = [], []
    for r in range(cp):
        kv_gid += [r, 2 * cp - 1 - r]
        kv_row += [2 * r * C, (2 * r + 1) * C]
    parts_k, parts_v = [], []
    for r in range(cp):
        geo = dict(chunk_len=C, q_chunk_gid=[r, 2 * cp - 1 - r], kv_chunk_gid=kv_gid, kv_chunk_row=kv_row)
        q_l, do_l = glue.zigzag_slice(q, cp, r).to(DEV), glue.zigzag_slice(d_o, cp, r).to(DEV)
        o, lse = ops.flash_attn(q_l, k_g, v_g, causal=True, return_lse=True, **geo)
        _, dk, dv = ops.flash_attn_bwd(q_l, k_g, v_g, o, do_l, lse, **geo)
        parts_k.append(dk[0].cpu()); parts_v.append(dv[0].cpu())               # [S (gathered order), Hkv, Dh] bf16

    def ring(parts):                         # segment p (rows of rank p): rank p+1 sends first, rank p adds last
        out = torch.empty_like(parts[0])
        for p_ in range(cp):
            seg = slice(2 * p_ * C, 2 * (p_ + 1) * C)
            acc = parts[(p_ + 1) % cp][seg]
            for h in range(2, cp + 1):
                acc = (acc.float() + parts[(p_ + h) % cp][seg].float()).bfloat16()
            out[seg] = acc
        return out

    def fp32_sum(parts):
        return sum(t.float() for t in parts).bfloat16()

    ref_k = torch.cat([glue.zigzag_slice(dk_r, cp, r) for r in range(cp)], 1)[0]
    ref_v = torch.cat([glue.zigzag_slice(dv_r, cp, r) for r in range(cp)], 1)[0]
    qd, kd, vd, dod = (t.to(DEV) for t in (q, k, v, d_o))
    o1, lse1 = ops.flash_attn(qd, kd, vd, causal=True, return_lse=True)
    _, dk1, dv1 = ops.flash_attn_bwd(qd, kd, vd, o1, dod, lse1)
    errs = {"dk_ring_bf16": rel_l2(ring(parts_k), ref_k), "dv_ring_bf16": rel_l2(ring(parts_v), ref_v),
            "dk_fp32_sum": rel_l2(fp32_sum(parts_k), ref_k), "dv_fp32_sum": rel_l2(fp32_sum(parts_v), ref_v),
            "dk_cp1": rel_l2(dk1, dk_r), "dv_cp1": rel_l2(dv1, dv_r)}
    record_parity("cp8_dkv_reduce_scatter_rounding_S4096", **errs)
    tol("ring dK", errs["dk_ring_bf16"], 5.8e-3)
    tol("ring dV", errs["dv_ring_bf16"], 5.4e-3)
    # the bf16 hops may cost at most a few bf16 roundings on top of the fp32-summed result
    assert errs["dk_ring_bf16"] < errs["dk_fp32_sum"] + 5e-3 and errs["dv_ring_bf16"] < errs["dv_fp32_sum"] + 5e-3


# ---------------------------------------------------------------------------------------------
# ViT training kernels (reference stage 2 trains the encoder)
# ---------------------------------------------------------------------------------------------
@pytest.mark.parametrize("rows,cols", [(70, 1024), (1025, 1024), (33, 4096)])
def test_layernorm_bwd(ops, rows, cols):
    x = (torch.randn(rows, cols, generator=g(40)) * 2 + 0.3).bfloat16()
    w = (1 + 0.1 * torch.randn(cols, generator=g(41))).bfloat16()
    b = (0.1 * torch.randn(cols, generator=g(42))).bfloat16()
    dy = torch.randn(rows, cols, generator=g(43)).bfloat16()
    xf, wf, bf = x.float().requires_grad_(True), w.float().requires_grad_(True), b.float().requires_grad_(True)
    torch.nn.functional.layer_norm(xf, (cols,), wf, bf, 1e-6).backward(dy.float())
    dg = torch.zeros(cols, dtype=torch.float32, device=DEV)
    db = torch.zeros_like(dg)
    dx = ops.layernorm_bwd(dy.to(DEV), x.to(DEV), w.to(DEV), 1e-6, dg, db)
    tol("dx", rel_l2(dx, xf.grad), 2.5e-3)                     # one bf16 rounding of the result
    tol("dgamma", rel_l2(dg, wf.grad), 1e-4)
    tol("dbeta", rel_l2(db, bf.grad), 1e-4)


def test_gelu_fwd_and_bias_scale_residual(ops):
    x = (torch.randn(300, 1024, generator=g(44)) * 2).bfloat16()
    ge, gr = ops.gelu(x.to(DEV)).cpu(), torch.nn.functional.gelu(x.float()).bfloat16()
    # device erff vs host erf: 1 + erf(x / sqrt 2) cancels for negative x — ~1 % of the results differ, by one bf16 ulp where the
    # value is O(1) and by ~1e-8 absolute in the far negative tail (|gelu| ~ 1e-6), where fp32 cancellation leaves few digits
    dgl = (ge.float() - gr.float()).abs()
    assert float((ge != gr).float().mean()) < 5e-2 and bool((dgl <= 8e-3 * gr.float().abs() + 1e-6).all())
    gt, gtr = ops.gelu(x.to(DEV), tanh=True).cpu(), torch.nn.functional.gelu(x.float(), approximate="tanh").bfloat16()
    dgt = (gt.float() - gtr.float()).abs()                                     # device tanhf vs host tanh
    assert float((gt != gtr).float().mean()) < 5e-2 and bool((dgt <= 8e-3 * gtr.float().abs() + 1e-6).all())
    rows, cols = 130, 1024
    y = torch.randn(rows, cols, generator=g(45)).bfloat16()
    bias = torch.randn(cols, generator=g(46)).bfloat16()
    ls = (0.1 + 0.02 * torch.randn(cols, generator=g(47))).bfloat16()
    res = torch.randn(rows, cols, generator=g(48)).bfloat16()
    go = torch.randn(rows, cols, generator=g(49)).bfloat16()
    for use_b, use_s in ((True, True), (True, False), (False, True)):
        yb, bb, sb, rb = (t.clone().requires_grad_(True) for t in (y, bias, ls, res))
        t = yb + bb if use_b else yb                          # the reference's three torch ops, bf16 (intern_vit_model.py:60-66)
        t = t * sb if use_s else t
        ref = rb + t
        ref.backward(go)
        out = ops.bias_scale_residual(y.to(DEV), bias.to(DEV) if use_b else None, ls.to(DEV) if use_s else None, res.to(DEV))
        assert torch.equal(out.cpu(), ref.detach())            # same rounding chain: bit-exact
        d_b = torch.zeros(cols, dtype=torch.float32, device=DEV) if use_b else None
        d_s = torch.zeros(cols, dtype=torch.float32, device=DEV) if use_s else None
        dx = ops.bias_scale_residual_bwd(go.to(DEV), y.to(DEV), bias.to(DEV) if use_b else None, ls.to(DEV) if use_s else None, d_b, d_s)
        assert torch.equal(dx.cpu(), yb.grad)
        if use_b:
            tol("d_bias", rel_l2(d_b, bb.grad), 6e-3)          # autograd sums bf16 rows in bf16 storage; the kernel in fp32
        if use_s:
            tol("d_scale", rel_l2(d_s, sb.grad), 6e-3)


@pytest.mark.parametrize("B,S,H,D", [(3, 1025, 16, 64), (2, 1024, 4, 64), (1, 300, 2, 128), (2, 729, 16, 96), (1, 1024, 3, 96), (1, 300, 2, 96)])
def test_non_causal_attention_backward_through_the_padded_chunk_tables(ops, B, S, H, D):
    """autograd_fns.FlashAttnNonCausalFn: the ViT's attention gradient from the causal backward kernels, un-masked through their chunk
    tables on zero-padded [S_pad, B * H, d] copies, vs fp32 autograd through the oracle (non-causal).  d = 96 (SigLIP's 72, padded): S_pad a
    multiple of 256 (729 -> 768, 1024) takes the pair kernel at d = 128 for dK + dV and the d = 96 general kernel on views of the same buffers
    for dQ; S_pad = 384 takes both general kernels at d = 96."""
    from long_vita_amd.autograd_fns import FlashAttnNonCausalFn
    q = torch.randn(B, S, H, D, generator=g(50)).bfloat16()
    k = torch.randn(B, S, H, D, generator=g(51)).bfloat16()
    v = torch.randn(B, S, H, D, generator=g(52)).bfloat16()
    d_o = torch.randn(B, S, H, D, generator=g(53)).bfloat16()
    qf, kf, vf = (t.float().requires_grad_(True) for t in (q, k, v))
    o = oattn.core_attention(qf.transpose(0, 1), kf.transpose(0, 1), vf.transpose(0, 1), False)      # [S, B, H*D]
    o.view(S, B, H, D).transpose(0, 1).backward(d_o.float())
    qd, kd, vd = (t.to(DEV).requires_grad_(True) for t in (q, k, v))
    out = FlashAttnNonCausalFn.apply(qd, kd, vd, 1.0 / math.sqrt(D))
    tol("out", rel_l2(out, o.detach().view(S, B, H, D).transpose(0, 1)), 3.4e-3)
    out.backward(d_o.to(DEV))
    tol("dq", rel_l2(qd.grad, qf.grad), 3.7e-3)
    tol("dk", rel_l2(kd.grad, kf.grad), 3.7e-3)
    tol("dv", rel_l2(vd.grad, vf.grad), 3.7e-3)
