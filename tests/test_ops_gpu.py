"""Parity of every HIP entry point (through the C ABI) against the CPU oracle on seeded inputs.
Integer / byte work is bit-exact; floating point tolerances are stated per test."""
import math

import pytest
import torch

pytestmark = pytest.mark.gpu

from oracle import attention as oattn  # noqa: E402
from oracle import glue  # noqa: E402
from oracle import vit as ovit  # noqa: E402

from conftest import tol  # noqa: E402

DEV = "cuda"


@pytest.fixture(scope="module")
def ops():
    from long_vita_amd import ops as _ops
    _ops._L.load(allow_build=False)     # fail loudly if the .so is missing on the GPU box
    return _ops


def g(seed):
    return torch.Generator().manual_seed(seed)


def rel_l2(a, b):
    a, b = a.float().cpu(), b.float().cpu()
    return float((a - b).norm() / (b.norm() + 1e-30))


def bf16_ulp_diff(a, b):
    """max difference in bf16 ulps (bit patterns as sign-magnitude integers)."""
    ai = a.cpu().view(torch.int16).to(torch.int32)
    bi = b.cpu().view(torch.int16).to(torch.int32)
    ai = torch.where(ai < 0, -(ai & 0x7FFF), ai)
    bi = torch.where(bi < 0, -(bi & 0x7FFF), bi)
    return int((ai - bi).abs().max())


# ---------------------------------------------------------------------------------------------
@pytest.mark.parametrize("rows,cols", [(1, 5120), (33, 5120), (257, 1024), (64, 4096), (5, 8192), (7, 1152)])
def test_rmsnorm(ops, rows, cols):
    x = (torch.randn(rows, cols, generator=g(1)) * 3).bfloat16()
    w = (1 + 0.1 * torch.randn(cols, generator=g(2))).bfloat16()
    ref = glue.rmsnorm(x, w, 1e-6)
    out = ops.rmsnorm(x.to(DEV), w.to(DEV), 1e-6)
    # fp32 statistic may differ in the last ulp (reduction order): the two chained bf16 roundings
    # (x*rstd, then *w) can each flip -> at most 2 bf16 ulps, and almost all elements bit-equal
    assert bf16_ulp_diff(out, ref) <= 2
    assert (out.cpu() == ref).float().mean() > 0.99


@pytest.mark.parametrize("rows,cols,eps", [(3, 1024, 1e-6), (130, 1024, 1e-6), (17, 4096, 1e-5), (9, 1152, 1e-6)])
def test_layernorm(ops, rows, cols, eps):
    x = (torch.randn(rows, cols, generator=g(3)) * 2 + 0.5).bfloat16()
    w = (1 + 0.1 * torch.randn(cols, generator=g(4))).bfloat16()
    b = (0.1 * torch.randn(cols, generator=g(5))).bfloat16()
    ref = torch.nn.functional.layer_norm(x.float(), (cols,), w.float(), b.float(), eps).bfloat16()
    out = ops.layernorm(x.to(DEV), w.to(DEV), b.to(DEV), eps)
    assert bf16_ulp_diff(out, ref) <= 1
    assert (out.cpu() == ref).float().mean() > 0.99


def _rope_ref(t, c2, s2):
    """`t * cos + rotate_half(t) * sin` in bf16 (apply_rotary_pos_emb_bshd, rotary_pos_embedding.py:200-204) spelled out as fp32 products
    rounded to bf16, summed in fp32, rounded — what torch's bf16 kernels compute, without depending on the host CPU's bf16 code path
    (one GPU box of r04 disagreed with the others in exactly these two bit-exact comparisons; its host ran torch's bf16 ops differently)."""
    a = (t.float() * c2.float()).bfloat16().float()
    b = (glue.rotate_half(t).float() * s2.float()).bfloat16().float()
    return (a + b).bfloat16()


def test_rope_table_and_apply(ops):
    inv = glue.rope_inv_freq(128, 1e6)
    assert torch.equal(ops.rope_inv_freq(128, 1e6, "cpu"), inv)
    pos = torch.cat([torch.arange(0, 300), torch.tensor([4095, 65535, 131071, 524288, 1048575])])
    cos, sin = ops.rope_table(pos.to(DEV), inv.to(DEV))
    freqs = torch.outer(pos.float(), inv)
    # device cosf/sinf vs torch CPU: <= 1 bf16 ulp after the cast
    assert bf16_ulp_diff(cos, torch.cos(freqs).bfloat16()) <= 1
    assert bf16_ulp_diff(sin, torch.sin(freqs).bfloat16()) <= 1
    # apply with the DEVICE's tables: the rounding chain bf16(bf16(t*cos)+bf16(rot*sin)) is bit-exact
    t = torch.randn(305, 6, 128, generator=g(6)).bfloat16()
    emb_cos, emb_sin = cos.cpu(), sin.cpu()
    c2 = torch.cat([emb_cos, emb_cos], -1)[:, None, :]
    s2 = torch.cat([emb_sin, emb_sin], -1)[:, None, :]
    ref = _rope_ref(t, c2, s2)
    out = ops.rope_apply_(t.to(DEV).clone(), cos, sin)
    assert torch.equal(out.cpu(), ref)
    # strided view + backward sign: R(-theta) R(theta) x ~= x
    big = torch.zeros(305, 8, 200, dtype=torch.bfloat16, device=DEV)
    view = big[:, 1:7, 8:136]
    view.copy_(t.to(DEV))
    ops.rope_apply_(view, cos, sin)
    assert torch.equal(view.cpu(), ref)
    assert float(big[:, 0].abs().max()) == 0 and float(big[:, :, :8].abs().max()) == 0


def test_rope_qkv_fused(ops):
    rows, ng, qpg, d = 70, 8, 5, 128
    mixed = torch.randn(rows, ng, (qpg + 2) * d, generator=g(7)).bfloat16()
    pos = torch.arange(1000, 1000 + rows)
    inv = glue.rope_inv_freq(d, 1e6)
    cos, sin = ops.rope_table(pos.to(DEV), inv.to(DEV))
    c2 = torch.cat([cos.cpu(), cos.cpu()], -1)[:, None, :]
    s2 = torch.cat([sin.cpu(), sin.cpu()], -1)[:, None, :]
    m = mixed.view(rows, ng, qpg + 2, d)
    ref = m.clone()
    for h in range(qpg + 1):
        ref[:, :, h] = _rope_ref(m[:, :, h], c2, s2)
    md = mixed.to(DEV).clone()
    kv = torch.empty(2, rows, ng, d, dtype=torch.bfloat16, device=DEV)
    ops.rope_qkv_(md, ng, qpg, d, cos, sin, kv)
    assert torch.equal(md.cpu().view(rows, ng, qpg + 2, d), ref)
    assert torch.equal(kv[0].cpu(), ref[:, :, qpg]) and torch.equal(kv[1].cpu(), ref[:, :, qpg + 1])
    # kv-head split packing: [split][2][rows][ng/split][d] (one all-gather message per split)
    for split in (2, 4):
        md2 = mixed.to(DEV).clone()
        kv2 = torch.empty(split, 2, rows, ng // split, d, dtype=torch.bfloat16, device=DEV)
        ops.rope_qkv_(md2, ng, qpg, d, cos, sin, kv2, split)
        hg = ng // split
        for j in range(split):
            assert torch.equal(kv2[j, 0].cpu(), ref[:, j * hg:(j + 1) * hg, qpg])
            assert torch.equal(kv2[j, 1].cpu(), ref[:, j * hg:(j + 1) * hg, qpg + 1])


# ---------------------------------------------------------------------------------------------
def test_row_gather_scatter_bit_exact(ops):
    table = torch.randn(1000, 5120, generator=g(8)).bfloat16()
    idx = torch.randint(0, 1000, (777,), generator=g(9))
    out = ops.row_gather(table.to(DEV), idx.to(DEV))
    assert torch.equal(out.cpu(), table[idx])
    with pytest.raises(IndexError):
        ops.row_gather(table.to(DEV), torch.tensor([0, 1000], device=DEV))
    # scatter forms of language_model_embedding.py:123,131
    we = torch.randn(2, 64, 5120, generator=g(10)).bfloat16()
    feats = torch.randn(3, 8, 5120, generator=g(11)).bfloat16()
    src_b = torch.tensor([0, 0, 2, 2, 2]); src_s = torch.tensor([1, 3, 0, 1, 7])
    tgt_b = torch.tensor([0, 0, 1, 1, 1]); tgt_s = torch.tensor([5, 6, 0, 1, 63])
    ref = we.clone(); ref[tgt_b, tgt_s] = feats[src_b, src_s]
    dst = we.to(DEV).clone().view(128, 5120)
    ops.row_scatter_(dst, (tgt_b * 64 + tgt_s).to(DEV), feats.to(DEV).view(24, 5120), (src_b * 8 + src_s).to(DEV))
    assert torch.equal(dst.cpu().view(2, 64, 5120), ref)
    # empty
    assert ops.row_gather(table.to(DEV), torch.empty(0, dtype=torch.int64, device=DEV)).shape == (0, 5120)
    # fp32 rows too
    t32 = torch.randn(50, 12, generator=g(12))
    assert torch.equal(ops.row_gather(t32.to(DEV), idx[:20].to(DEV) % 50).cpu(), t32[idx[:20] % 50])


@pytest.mark.parametrize("n,p", [(1, 1.0), (1000, 0.01), (131072, 0.004), (131072, 0.5), (5, 0.0), (1048576, 1e-5)])
def test_mask_to_index(ops, n, p):
    mask = torch.rand(n, generator=g(13)) < p
    out = ops.mask_to_index(mask.to(DEV))
    assert torch.equal(out.cpu(), mask.nonzero().flatten())


@pytest.mark.parametrize("name", ["survey_appendix", "video_like", "straddle"])
def test_cp_batch_slice_matches_reference_fixture(ops, name):
    """get_batch_on_this_cp_rank on the device vs the fixture produced by the reference's own code."""
    from conftest import load_golden
    from long_vita_amd.training_utils import get_batch_on_this_cp_rank

    case = [c for c in load_golden("cp_slice.pt")["cases"] if c["name"] == name][0]
    for r in range(case["cp"]):
        batch = {"tokens": case["tokens"].to(DEV), "labels": (case["tokens"] + 1).to(DEV),
                 "position_ids": torch.arange(case["seq"])[None].to(DEV),
                 "external_images": case["images"].to(DEV), "external_indices": case["indices"].to(DEV)}
        mine = get_batch_on_this_cp_rank(batch, seq_length=case["seq"], cp_size=case["cp"], cp_rank=r)
        ref = case["out"][r]
        assert set(mine.keys()) == set(ref.keys())
        for k in ref:
            assert mine[k].dtype == ref[k].dtype and torch.equal(mine[k].cpu(), ref[k]), (name, r, k)


# ---------------------------------------------------------------------------------------------
def _gemm_ref(a, w, epi, bias, scale, res, ops):
    v = a.float() @ w.float().t()
    bf = lambda t: t.bfloat16().float()
    if epi == ops.EPI_NONE:
        return v.bfloat16()
    if epi == ops.EPI_BIAS:
        return (v + bias.float()).bfloat16()
    if epi == ops.EPI_BIAS_GELU:
        t = bf(v + (bias.float() if bias is not None else 0))
        return torch.nn.functional.gelu(t).bfloat16()
    if epi == ops.EPI_RESIDUAL:
        return (res.float() + bf(v + (bias.float() if bias is not None else 0))).bfloat16()
    if epi == ops.EPI_BIAS_SCALE_RES:
        return (res.float() + bf(bf(v + bias.float()) * scale.float())).bfloat16()
    if epi == ops.EPI_SWIGLU:
        gate, up = torch.chunk(v, 2, dim=-1)
        gte, u = bf(gate), bf(up)
        return (bf(torch.nn.functional.silu(gte)) * u).bfloat16()
    raise AssertionError


@pytest.mark.parametrize("M,N,K", [(128, 128, 64), (256, 384, 1024), (1025, 1024, 1024), (300, 200, 640),
                                   (77, 5120, 1024), (2048, 7168, 5120)])
@pytest.mark.parametrize("epi", [0, 1, 2, 3, 4, 5])
@pytest.mark.parametrize("kernel", ["auto", "w4", "w8"])
def test_gemm_epilogues(ops, M, N, K, epi, kernel, monkeypatch):
    """kernel: "auto" = the library's own choice (small tile below 192 big tiles, the 4-wave AGPR kernel above), "w4" / "w8" force
    the large-problem kernels (4 waves x 16x16x32 with pinned accumulators / 8 waves x 32x32x16) onto every shape, so that their
    ragged-edge clamping, single-K-tile prologue and every epilogue are covered at small sizes too."""
    if epi == 5 and N % 2:
        pytest.skip("swiglu needs even rows")
    if M * N * K > 2e10 and epi not in (0, 5) and kernel != "w4":
        pytest.skip("large shape only for plain / swiglu")
    if kernel != "auto":
        monkeypatch.setenv("VITA_GEMM_KERNEL", kernel)
    a = (torch.randn(M, K, generator=g(20)) * 0.5).bfloat16()
    wrows = 2 * N if epi == ops.EPI_SWIGLU else N
    w = (torch.randn(wrows, K, generator=g(21)) * (1.0 / math.sqrt(K))).bfloat16()
    bias = (torch.randn(N, generator=g(22)) * 0.1).bfloat16() if epi in (1, 2, 3, 4) else None
    scale = (0.1 + 0.01 * torch.randn(N, generator=g(23))).bfloat16() if epi == 4 else None
    res = torch.randn(M, N, generator=g(24)).bfloat16() if epi in (3, 4) else None
    ref = _gemm_ref(a, w, epi, bias, scale, res, ops)
    out = ops.gemm(a.to(DEV), w.to(DEV), epi, None if bias is None else bias.to(DEV),
                   None if scale is None else scale.to(DEV), None if res is None else res.to(DEV))
    # fp32 accumulation order differs from the CPU GEMM -> a few results flip one bf16 ulp
    err = rel_l2(out, ref)
    assert err < 2e-3, err
    assert bf16_ulp_diff(out, ref) <= 2 or (out.cpu().float() - ref.float()).abs().max() < 2e-2


def test_gemm_asymmetric_identity(ops):
    """A = I with an asymmetric W catches a transposed C write (cdna guide §3)."""
    K = 128
    a = torch.eye(K).bfloat16()
    w = (torch.arange(192 * K).reshape(192, K) % 251 - 125).float().bfloat16()
    out = ops.gemm(a.to(DEV), w.to(DEV))
    assert torch.equal(out.cpu(), w.t().contiguous())


def test_gemm_strided_a_and_errors(ops):
    big = torch.randn(100, 2048, generator=g(25)).bfloat16().to(DEV)
    a = big[:, 512:512 + 1024]
    w = torch.randn(64, 1024, generator=g(26)).bfloat16().to(DEV)
    ref = (a.float().cpu() @ w.float().cpu().t()).bfloat16()
    tol("ops.gemm(a, w), ref", rel_l2(ops.gemm(a, w), ref), 2e-3)
    with pytest.raises(RuntimeError):
        ops.gemm(a, torch.zeros(64, 512, dtype=torch.bfloat16, device=DEV))     # weight shape mismatch
    # a contraction that is not a multiple of 64 (SigLIP's FFN 4304) runs on zero-padded copies: exact
    a100, w100 = big[:, :100].contiguous(), torch.randn(8, 100, generator=g(29)).bfloat16().to(DEV)
    ref100 = (a100.float().cpu() @ w100.float().cpu().t()).bfloat16()
    tol("ops.gemm(a100, w100), ref100", rel_l2(ops.gemm(a100, w100), ref100), 2e-3)
    with pytest.raises(RuntimeError):                                            # the C entry point itself still refuses it
        ops._L.check(ops._L.load().vita_gemm_bf16(a100.data_ptr(), 100, w100.data_ptr(), 100, torch.empty(100, 8, dtype=torch.bfloat16,
                     device=DEV).data_ptr(), 8, 100, 8, 100, 0, None, None, None, 0, None), "vita_gemm_bf16")
    with pytest.raises(RuntimeError):
        ops.rmsnorm(torch.zeros(4, 64).bfloat16(), torch.ones(64).bfloat16())    # CPU tensor: no fallback


@pytest.mark.parametrize("M", [1, 2, 5, 16])
def test_gemm_skinny(ops, M):
    K, N = 5120, 3000
    a = torch.randn(M, K, generator=g(27)).bfloat16()
    w = (torch.randn(N, K, generator=g(28)) * 0.02).bfloat16()
    ref = a.float() @ w.float().t()
    out = ops.gemm_skinny(a.to(DEV), w.to(DEV), out_f32=True)
    torch.testing.assert_close(out.cpu(), ref, rtol=1e-4, atol=1e-4)
    outb = ops.gemm_skinny(a.to(DEV), w.to(DEV))
    tol("outb, ref", rel_l2(outb, ref), 2.5e-03)


# ---------------------------------------------------------------------------------------------
def _attn_ref(q, k, v, causal, q_pos=None, k_pos=None):
    # q [B,S,H,D] -> oracle layout [s,b,h,d]
    o = oattn.core_attention(q.transpose(0, 1).float(), k.transpose(0, 1).float(), v.transpose(0, 1).float(), causal,
                             q_pos=q_pos, k_pos=k_pos)
    B, S, H, D = q.shape
    return o.view(S, B, H, D).transpose(0, 1)


@pytest.mark.parametrize("S,Hq,Hkv,D,causal", [
    (256, 5, 1, 128, True), (512, 10, 2, 128, True), (1024, 40, 8, 128, True), (320, 5, 1, 128, True),
    (1025, 16, 16, 64, False), (192, 4, 4, 64, False), (2048, 5, 1, 128, True), (704, 2, 2, 128, False),
    # r05: head size 96 (SigLIP's 72 zero-padded): 12-slot rows, rotated LDS layout; ragged rows and keys, GQA, causal
    (729, 16, 16, 96, False), (1024, 4, 4, 96, False), (333, 6, 2, 96, False), (640, 4, 2, 96, True), (61, 2, 2, 96, False),
])
def test_flash_attention_single_chunk(ops, S, Hq, Hkv, D, causal):
    B = 2 if D in (64, 96) else 1
    q = torch.randn(B, S, Hq, D, generator=g(30)).bfloat16()
    k = torch.randn(B, S, Hkv, D, generator=g(31)).bfloat16()
    v = torch.randn(B, S, Hkv, D, generator=g(32)).bfloat16()
    ref = _attn_ref(q, k, v, causal)
    out, lse = ops.flash_attn(q.to(DEV), k.to(DEV), v.to(DEV), causal=causal, return_lse=True)
    # P is rounded to bf16 before PV (as flash-attn / TE do): tolerance 1e-2 relative L2, 3e-2 abs
    tol("out, ref", rel_l2(out, ref), 3.4e-03)
    assert float((out.cpu().float() - ref).abs().max()) < 3e-2
    # lse against fp32 math
    sc = torch.einsum("bqhd,bkhd->bhqk", q.float(), k.float().repeat_interleave(Hq // Hkv, 2)) / math.sqrt(D)
    if causal:
        sc = sc.masked_fill(torch.triu(torch.ones(S, S, dtype=torch.bool), 1), float("-inf"))
    torch.testing.assert_close(lse.cpu(), torch.logsumexp(sc, -1), rtol=2e-3, atol=2e-3)


def test_flash_attention_forced_rescale(ops):
    """Spike one key so the running max jumps late in the sequence (cdna guide rule 26)."""
    S, D = 1024, 128
    q = torch.randn(1, S, 5, D, generator=g(33)).bfloat16()
    k = torch.randn(1, S, 1, D, generator=g(34)).bfloat16()
    v = torch.randn(1, S, 1, D, generator=g(35)).bfloat16()
    k[0, 700] = (q[0, 900, 2].float() * 3).bfloat16()       # huge score for (row 900, key 700)
    ref = _attn_ref(q, k, v, True)
    out = ops.flash_attn(q.to(DEV), k.to(DEV), v.to(DEV), causal=True)
    tol("out, ref", rel_l2(out, ref), 3.0e-03)
    assert float((out.cpu().float() - ref).abs().max()) < 5e-2


def test_flash_attention_mixed_qkv_views(ops):
    """Q/K/V read in place from Megatron's mixed QKV activation [S, ng, (qpg+2), d] (grouped q view)."""
    S, ng, qpg, d = 512, 8, 5, 128
    mixed = torch.randn(1, S, ng, qpg + 2, d, generator=g(36)).bfloat16().to(DEV)
    q5 = mixed[:, :, :, :qpg]                  # [1, S, ng, qpg, d]  group stride (qpg+2)*d
    kview = mixed[:, :, :, qpg]                # [1, S, ng, d]
    vview = mixed[:, :, :, qpg + 1]
    out = ops.flash_attn(q5, kview, vview, causal=True)
    ref = _attn_ref(q5.reshape(1, S, ng * qpg, d).cpu(), kview.cpu(), vview.cpu(), True)
    tol("out, ref", rel_l2(out, ref), 3.1e-03)


@pytest.mark.parametrize("cp,S", [(2, 2048), (4, 4096), (8, 4096)])
def test_flash_attention_zigzag_chunks(ops, cp, S):
    """Every rank's zig-zag context-parallel attention (local Q x all-gathered K/V in rank order)
    re-assembled == monolithic causal attention (SURVEY.md §8c cross-check ii)."""
    Hq, Hkv, D = 5, 1, 128
    C = S // (2 * cp)
    q = torch.randn(1, S, Hq, D, generator=g(40)).bfloat16()
    k = torch.randn(1, S, Hkv, D, generator=g(41)).bfloat16()
    v = torch.randn(1, S, Hkv, D, generator=g(42)).bfloat16()
    full = _attn_ref(q, k, v, True)
    # gathered buffer = concat over ranks of each rank's local (zig-zag) K/V
    k_g = torch.cat([glue.zigzag_slice(k, cp, r) for r in range(cp)], 1).to(DEV)
    v_g = torch.cat([glue.zigzag_slice(v, cp, r) for r in range(cp)], 1).to(DEV)
    kv_gid, kv_row = [], []
    for r in range(cp):
        kv_gid += [r, 2 * cp - 1 - r]
        kv_row += [2 * r * C, (2 * r + 1) * C]
    for r in range(cp):
        q_l = glue.zigzag_slice(q, cp, r).to(DEV)
        out = ops.flash_attn(q_l, k_g, v_g, causal=True, chunk_len=C, q_chunk_gid=[r, 2 * cp - 1 - r],
                             kv_chunk_gid=kv_gid, kv_chunk_row=kv_row)
        ref = glue.zigzag_slice(full, cp, r)
        tol("out, ref", rel_l2(out, ref), 3.5e-03)
        assert float((out.cpu().float() - ref).abs().max()) < 3e-2


# ---------------------------------------------------------------------------------------------
def test_vit_front_back_kernels(ops):
    cfg = ovit.ViTConfig(num_layers=0)
    p = ovit.init_vit_params(cfg, seed=5)
    images = torch.randn(2, 3, 448, 448, generator=g(50)).bfloat16()
    # patchify is pure data movement: bit-exact against unfold
    patches = ops.patchify14(images.to(DEV), 640)
    ref = torch.nn.functional.unfold(images.float(), 14, stride=14).transpose(1, 2).reshape(-1, 588).bfloat16()
    assert torch.equal(patches[:, :588].cpu(), ref) and float(patches[:, 588:].abs().max()) == 0
    # patch GEMM + assemble == conv + cls + pos (oracle vit_embed)
    w = torch.zeros(1024, 640, dtype=torch.bfloat16); w[:, :588] = p["conv_w"].reshape(1024, 588)
    pe = ops.gemm(patches, w.to(DEV), ops.EPI_BIAS, p["conv_b"].to(DEV))
    x = ops.vit_assemble(pe, p["cls"].to(DEV).view(-1), p["pos"].to(DEV), 2, 1024)
    xr = ovit.vit_embed(images, p, cfg)
    tol("x, xr", rel_l2(x, xr), 4.4e-05)
    # pixel-shuffle + LayerNorm
    y = ops.pixel_shuffle_ln(xr.to(DEV), p["proj_ln_w"].to(DEV), p["proj_ln_b"].to(DEV), 32, True, 1e-5)
    t = glue.pixel_shuffle(xr[:, 1:].reshape(2, 32, 32, -1), 0.5).reshape(2, 256, 4096)
    yr = torch.nn.functional.layer_norm(t.float(), (4096,), p["proj_ln_w"].float(), p["proj_ln_b"].float(), 1e-5).bfloat16()
    assert bf16_ulp_diff(y, yr) <= 1


# ---------------------------------------------------------------------------------------------
# packed sequences: block-diagonal causal attention, forward and backward
# ---------------------------------------------------------------------------------------------
# S % 256 == 0: the 64-row kernels' packed variants (attn64.hip / attn_bwd64.hip / attn_bwd_kv64.hip, r03); otherwise attn.hip / attn_bwd.hip
@pytest.mark.parametrize("S,cu", [(1024, [0, 300, 301, 777, 1024]), (2048, [0, 2048]), (1536, [0, 64, 128, 900]),
                                  (512, [0, 255, 256, 257, 512]), (4096, [0, 1, 70, 1400, 1408, 3333]), (896, [0, 100, 640]),
                                  (2048, [0, 256, 512, 1024, 1280])])
def test_flash_attn_packed_sequences_fwd_bwd(ops, S, cu):
    from oracle import attention as oattn
    Hq, Hkv, D = 10, 2, 128
    g = torch.Generator().manual_seed(S + len(cu))
    q = (torch.randn(S, 1, Hq, D, generator=g) * 0.5).bfloat16()
    k = (torch.randn(S, 1, Hkv, D, generator=g) * 0.5).bfloat16()
    v = (torch.randn(S, 1, Hkv, D, generator=g) * 0.5).bfloat16()
    d_o = (torch.randn(S, 1, Hq * D, generator=g) * 0.1).bfloat16()
    cu_t = torch.tensor(cu, dtype=torch.int32)
    cu_full = cu_t if cu[-1] == S else torch.cat([cu_t, torch.tensor([S], dtype=torch.int32)])   # tail = its own sample
    qf, kf, vf = (t.float().requires_grad_(True) for t in (q, k, v))
    ref = oattn.core_attention(qf, kf, vf, causal=True, cu_seqlens=cu_full)
    ref.backward(d_o.float())
    seg_start, seg_end = ops.segments_from_cu_seqlens(cu_t.to(DEV), S)
    assert seg_start.tolist() == [max(c for c in cu_full.tolist() if c <= i) for i in range(S)]
    qd, kd, vd = (t.permute(1, 0, 2, 3).contiguous().to(DEV) for t in (q, k, v))          # [1, S, H, D]
    out, lse = ops.flash_attn(qd, kd, vd, causal=True, return_lse=True, seg_start=seg_start)
    tol("out.reshape(S, -1), ref.detach().reshape(S, -1)", rel_l2(out.reshape(S, -1), ref.detach().reshape(S, -1)), 3.2e-03)
    # rows that start a sample attend to themselves only: output == their own V (GQA: head h uses kv head h // 5)
    for r in [c for c in cu_full.tolist()[:-1]]:
        want = vd[0, r].repeat_interleave(Hq // Hkv, dim=0)
        tol("out[0, r], want", rel_l2(out[0, r], want), 1e-2)
    if S % 128 == 0:
        dod = d_o.view(S, Hq, D)[None].contiguous().to(DEV)
        dq, dk, dv = ops.flash_attn_bwd(qd, kd, vd, out, dod, lse, seg_start=seg_start, seg_end=seg_end)
        tol("dq[0], qf.grad[:, 0]", rel_l2(dq[0], qf.grad[:, 0]), 3.8e-03)
        tol("dk[0], kf.grad[:, 0]", rel_l2(dk[0], kf.grad[:, 0]), 3.8e-03)
        tol("dv[0], vf.grad[:, 0]", rel_l2(dv[0], vf.grad[:, 0]), 3.5e-03)


def test_gemm_siglip_epilogues(ops):
    """BIAS2_* epilogues: the bias meets the bf16-ROUNDED product (Megatron skip_bias_add), then tanh-GELU / residual."""
    g = torch.Generator().manual_seed(5)
    M, N, K = 300, 4352, 1152
    a = (torch.randn(M, K, generator=g) * 0.5).bfloat16()
    w = (torch.randn(N, K, generator=g) * 0.05).bfloat16()
    b = torch.randn(N, generator=g).bfloat16()
    r = torch.randn(M, N, generator=g).bfloat16()
    prod = (a.float() @ w.float().t()).bfloat16()
    pre = (prod + b)                                                         # bf16 add
    ref_act = torch.nn.functional.gelu(pre.float(), approximate="tanh").bfloat16()
    ref_res = r + pre
    ad, wd, bd, rd = a.to(DEV), w.to(DEV), b.to(DEV), r.to(DEV)
    out_act = ops.gemm(ad, wd, ops.EPI_BIAS2_GELU_TANH, bd).cpu()
    out_res = ops.gemm(ad, wd, ops.EPI_BIAS2_RES, bd, residual=rd).cpu()
    for out, ref in ((out_act, ref_act), (out_res, ref_res)):
        d = (out.float() - ref.float()).abs()
        assert float((d == 0).float().mean()) > 0.97                         # accumulation order moves a few roundings
        assert float((d / (ref.float().abs() + 1.0)).max()) < 2e-2
    with pytest.raises(ValueError):
        ops.gemm(ad, wd, ops.EPI_BIAS2_RES, bd)                              # residual required


# ---------------------------------------------------------------------------------------------
@pytest.mark.parametrize("scale,cap", [(0.5, 0.0), (0.0, 30.0), (1.7, 20.0)])
def test_logit_scale_and_softcap(ops, scale, cap):
    """args.output_multiplier_scale / output_logit_softcapping (M/core/models/multimodal/gpt_vl_model.py:349-355): the
    reference's bf16 op chain; the backward factor against torch autograd of the same expression in fp32."""
    x = (torch.randn(5, 152064, generator=g(71)) * 12).bfloat16()
    ref = glue.logit_postprocess(x, scale or None, cap or None)
    out = ops.logit_postprocess_(x.to(DEV).clone(), scale, cap).cpu()
    d = (out.view(torch.int16).int() - ref.view(torch.int16).int()).abs()          # device tanhf vs host tanh: <= 1 bf16 ulp
    assert int(d.max()) <= 1 and float((d > 0).float().mean()) < 2e-2
    xf = x.float().requires_grad_(True)
    glue.logit_postprocess(xf, scale or None, cap or None).backward(torch.ones_like(xf))
    gr = torch.ones(5, 152064).bfloat16().to(DEV)
    ops.logit_postprocess_bwd_(out.to(DEV), gr, scale, cap)
    tol("gr, xf.grad", rel_l2(gr, xf.grad), 6.2e-03)


# ---------------------------------------------------------------------------------------------
def test_cp_attention_through_the_c_abi_on_one_real_rccl_rank(ops):
    """vita_cp_unique_id / vita_cp_init / vita_cp_attn_fwd / vita_cp_attn_bwd / vita_cp_destroy (include/vita_hip.h) driven
    with raw pointers the way a non-Python host would: an RCCL communicator of ONE rank (what a 1-GPU box can hold), the K/V
    all-gathers per kv-head split on the library's communication stream, chunk-table attention, dK/dV reduce-scatter —
    equal to the single-device kernels on the same data (forward bit for bit: same kernel, same geometry)."""
    import ctypes as C
    from long_vita_amd import lib as L
    h = L.load()
    S, ng, qpg, d, n_split = 1024, 4, 2, 128, 2
    hq = ng * qpg
    mixed = torch.randn(1, S, ng, qpg + 2, d, generator=g(81)).bfloat16().to(DEV)
    q5, k, v = mixed[:, :, :, :qpg], mixed[:, :, :, qpg], mixed[:, :, :, qpg + 1]
    hg = ng // n_split
    kv_packed = torch.stack([torch.stack([k[0, :, j * hg:(j + 1) * hg], v[0, :, j * hg:(j + 1) * hg]]) for j in range(n_split)]).contiguous()
    uid = C.create_string_buffer(128)
    L.check(h.vita_cp_unique_id(uid), "vita_cp_unique_id")
    ctx = C.c_void_p()
    L.check(h.vita_cp_init(C.byref(ctx), 1, 0, uid), "vita_cp_init")
    try:
        nbytes = h.vita_cp_attn_workspace_bytes(1, S, ng, d)
        assert nbytes == 2 * S * ng * d * 2
        ws = torch.empty(nbytes, dtype=torch.uint8, device=DEV)
        dws = torch.empty(nbytes, dtype=torch.uint8, device=DEV)
        out = torch.empty(S, hq, d, dtype=torch.bfloat16, device=DEV)
        lse = torch.empty(hq, S, dtype=torch.float32, device=DEV)
        p = L.CpAttnParams()
        p.q, p.q_row_stride, p.q_head_stride, p.q_group_stride = q5.data_ptr(), q5.stride(1), q5.stride(3), q5.stride(2)
        p.kv_packed = kv_packed.data_ptr()
        p.out, p.out_row_stride, p.out_head_stride, p.lse = out.data_ptr(), out.stride(0), out.stride(1), lse.data_ptr()
        p.s_local, p.n_q_heads, p.n_kv_heads, p.head_dim, p.n_split = S, hq, ng, d, n_split
        p.softmax_scale = 1.0 / math.sqrt(d)
        p.workspace, p.workspace_bytes, p.dkv_workspace = ws.data_ptr(), nbytes, dws.data_ptr()
        st = torch.cuda.current_stream().cuda_stream
        L.check(h.vita_cp_attn_fwd(ctx, C.byref(p), st), "vita_cp_attn_fwd")
        torch.cuda.synchronize()
        ref, ref_lse = ops.flash_attn(q5, k, v, causal=True, return_lse=True)
        assert torch.equal(out, ref[0]) and torch.equal(lse, ref_lse[0])
        # backward
        d_o = torch.randn(1, S, hq, d, generator=g(82)).bfloat16().to(DEV)
        dq_r, dk_r, dv_r = ops.flash_attn_bwd(q5, k.contiguous().view(1, S, ng, d), v.contiguous().view(1, S, ng, d), ref, d_o, ref_lse)
        delta = torch.empty(hq, S, dtype=torch.float32, device=DEV)
        L.check(h.vita_attn_delta(out.data_ptr(), d_o.data_ptr(), delta.data_ptr(), S, hq, d, out.stride(0), out.stride(1), d_o.stride(1),
                                  d_o.stride(2), st), "vita_attn_delta")
        dmixed = torch.zeros_like(mixed)
        dq5 = dmixed[:, :, :, :qpg]
        dkv = torch.empty_like(kv_packed)
        L.check(h.vita_cp_attn_bwd(ctx, C.byref(p), d_o.data_ptr(), lse.data_ptr(), delta.data_ptr(), dq5.data_ptr(), dkv.data_ptr(), st),
                "vita_cp_attn_bwd")
        torch.cuda.synchronize()
        tol("dq5, dq_r", rel_l2(dq5, dq_r), 1e-3)
        dk_c = torch.cat([dkv[j, 0] for j in range(n_split)], 1)
        dv_c = torch.cat([dkv[j, 1] for j in range(n_split)], 1)
        assert rel_l2(dk_c, dk_r[0]) < 1e-3 and rel_l2(dv_c, dv_r[0]) < 1e-3
        bad = L.CpAttnParams()
        assert h.vita_cp_attn_fwd(ctx, C.byref(bad), st) == L.VITA_ERR_INVALID_ARG
    finally:
        L.check(h.vita_cp_destroy(ctx), "vita_cp_destroy")


def test_attention_over_own_chunks_then_remote_chunks_merged_equals_one_launch(ops):
    """What dot_product_attention.forward_cp does for the first kv-head split (the rank's own zig-zag chunks are attended
    to while the K/V all-gather is in flight): a launch over the own chunks, a launch over the remote chunks only — a geometry without
    a diagonal chunk, where the rows of the rank's FIRST chunk may see nothing at all (lse = -inf) — and vita_attn_merge of the two
    partials == one launch over every chunk (and == the fp32 oracle)."""
    cp, hg, G, D, C = 4, 2, 5, 128, 512
    s_l, S = 2 * C, 2 * cp * C
    for r in (0, 2, 3):                                      # rank 0: its chunk 0 sees no remote key
        q = (torch.randn(1, s_l, hg * G, D, generator=g(300 + r)) * 0.5).bfloat16().to(DEV)
        rows = torch.randn(cp * 2 * s_l, hg, D, generator=g(310 + r)).bfloat16().to(DEV)       # [rank p][K | V][s_l][hg][d]
        own = [r, 2 * cp - 1 - r]
        kv_gid, kv_row = [], []
        for p in range(cp):
            kv_gid += [p, 2 * cp - 1 - p]
            kv_row += [p * 2 * s_l, p * 2 * s_l + C]
        k_all, v_all = rows.unsqueeze(0), rows[s_l:].unsqueeze(0)
        whole, lse_w = ops.flash_attn(q, k_all, v_all, causal=True, chunk_len=C, q_chunk_gid=own, kv_chunk_gid=kv_gid,
                                      kv_chunk_row=kv_row, return_lse=True)
        loc = [i for i in range(2 * cp) if i // 2 == r]
        rem = [i for i in range(2 * cp) if i // 2 != r]
        o_a, lse_a = ops.flash_attn(q, k_all, v_all, causal=True, chunk_len=C, q_chunk_gid=own, kv_chunk_gid=[kv_gid[i] for i in loc],
                                    kv_chunk_row=[kv_row[i] for i in loc], return_lse=True)
        o_b, lse_b = ops.flash_attn(q, k_all, v_all, causal=True, chunk_len=C, q_chunk_gid=own, kv_chunk_gid=[kv_gid[i] for i in rem],
                                    kv_chunk_row=[kv_row[i] for i in rem], return_lse=True)
        if r == 0:
            assert torch.isinf(lse_b[0, :, :C]).all() and float(o_b[0, :C].abs().max()) == 0.0       # nothing remote is visible
        ops.attn_merge_(o_a, lse_a, o_b, lse_b)
        tol("o_a, whole", rel_l2(o_a, whole), 4e-3)
        assert float((lse_a - lse_w).abs().max()) < 2e-3
        assert torch.isfinite(o_a.float()).all()


def test_cp_attention_c_abi_own_chunks_first_with_an_external_exchange(ops):
    """vita_cp_attn_fwd with `scratch` (ABI 13): split 0 attends to the rank's OWN zig-zag chunks straight from its packed shard
    (while gather 0 would be in flight), then to the remote chunks, and merges — the C twin of forward_cp's own-chunks-first
    (VERDICT r2 "missing" 6).  Driven through a context WITHOUT a communicator (vita_cp_init(..., unique_id = NULL): the host
    performs the exchange — here one process lays out the gathered K / V of CP = 4 ranks by hand), for ranks 0 (its first chunk
    sees no remote key), 2 and 3: == the same call without scratch (one launch per split) == the fp32 oracle; the rank's own slot
    of the gathered workspace is poisoned to prove split 0's own-chunk launch never reads it... (it is read by the remote launch of
    no one: own chunks are excluded from the remote tables)."""
    import ctypes as C
    from long_vita_amd import lib as L
    h = L.load()
    cp, ng, qpg, d, n_split, c = 4, 4, 2, 128, 2, 512
    s_l, S, hq, hg = 2 * c, 2 * cp * c, ng * qpg, ng // n_split
    k_full = (torch.randn(S, ng, d, generator=g(91)) * 0.5).bfloat16()
    v_full = torch.randn(S, ng, d, generator=g(92)).bfloat16()
    q_full = (torch.randn(S, hq, d, generator=g(93)) * 0.5).bfloat16()
    ref = _attn_ref(q_full[None], k_full[None], v_full[None], True)[0]                           # [S, hq, d] fp32
    pos = [glue.calibration_index(S, cp, r) for r in range(cp)]
    for r in (0, 2, 3):
        ctx = C.c_void_p()
        L.check(h.vita_cp_init(C.byref(ctx), cp, r, None), "vita_cp_init (external exchange)")
        try:
            q = q_full[pos[r]].to(DEV).contiguous()                                                 # [s_l, hq, d]
            q5 = q.view(1, s_l, ng, qpg, d)
            # packed shards [n_split][2][s_l][hg][d] of every rank; the gathered workspace is [n_split][cp][2][s_l][hg][d]
            packed = [torch.stack([torch.stack([k_full[pos[p_]][:, j * hg:(j + 1) * hg], v_full[pos[p_]][:, j * hg:(j + 1) * hg]])
                                   for j in range(n_split)]).to(DEV).contiguous() for p_ in range(cp)]
            ws = torch.stack([torch.stack([packed[p_][j] for p_ in range(cp)]) for j in range(n_split)]).contiguous()
            outs = {}
            for mode in ("single", "own_first"):
                w = ws.clone()
                if mode == "own_first":
                    w[0, r] = float("nan")                       # split 0: the rank's own slot is never read (own chunks come from kv_packed)
                out = torch.empty(s_l, hq, d, dtype=torch.bfloat16, device=DEV)
                lse = torch.empty(hq, s_l, dtype=torch.float32, device=DEV)
                nsc = h.vita_cp_attn_scratch_bytes(s_l, hq, n_split, d)
                assert nsc >= s_l * (hq // n_split) * d * 2 + 2 * (hq // n_split) * s_l * 4
                scratch = torch.empty(nsc, dtype=torch.uint8, device=DEV)
                p = L.CpAttnParams()
                p.q, p.q_row_stride, p.q_head_stride, p.q_group_stride = q5.data_ptr(), q5.stride(1), q5.stride(3), q5.stride(2)
                p.kv_packed = packed[r].data_ptr()
                p.out, p.out_row_stride, p.out_head_stride, p.lse = out.data_ptr(), out.stride(0), out.stride(1), lse.data_ptr()
                p.s_local, p.n_q_heads, p.n_kv_heads, p.head_dim, p.n_split = s_l, hq, ng, d, n_split
                p.softmax_scale = 1.0 / math.sqrt(d)
                p.workspace, p.workspace_bytes = w.data_ptr(), w.numel() * 2
                if mode == "own_first":
                    p.scratch, p.scratch_bytes = scratch.data_ptr(), nsc
                L.check(h.vita_cp_attn_fwd(ctx, C.byref(p), torch.cuda.current_stream().cuda_stream), "vita_cp_attn_fwd")
                torch.cuda.synchronize()
                outs[mode] = (out.clone(), lse.clone())
            assert torch.isfinite(outs["own_first"][0].float()).all()
            tol("own-chunks-first vs one launch per split", rel_l2(outs["own_first"][0], outs["single"][0]), 3.3e-3)   # two bf16 roundings instead of one
            assert float((outs["own_first"][1] - outs["single"][1]).abs().max()) < 2e-3
            # split 1 takes the plain path in both modes: bit-identical
            assert torch.equal(outs["own_first"][0][:, hq // n_split:], outs["single"][0][:, hq // n_split:])
            tol("vs the fp32 oracle", rel_l2(outs["own_first"][0], ref[pos[r]]), 4.1e-3)
        finally:
            L.check(h.vita_cp_destroy(ctx), "vita_cp_destroy")


@pytest.mark.parametrize("M,N,K", [(256, 256, 64), (256, 256, 128), (512, 768, 1024), (1536, 1024, 2112), (7168, 5120, 4096)])
def test_gemm_tn_both_operands_contraction_major(ops, M, N, K):
    """vita_gemm_bf16_tn: C[M, N] = A_t[K, M]^T W_t[K, N] (the wgrad GEMM, grad_output.t().matmul(total_input),
    M/core/tensor_parallel/layers.py:522-523, without transposed copies) vs fp32 math, vs the NT kernel fed with vita_transpose_bf16
    copies (same MFMA order along the contraction: bit-identical), and an asymmetric-identity case that catches swapped roles."""
    a_t = (torch.randn(K, M, generator=g(100)) * 0.5).bfloat16()
    w_t = (torch.randn(K, N, generator=g(101)) * (1.0 / math.sqrt(K))).bfloat16()
    ref = (a_t.float().t() @ w_t.float()).bfloat16()
    ad, wd = a_t.to(DEV), w_t.to(DEV)
    assert ops.gemm_tn_ok(ad, wd)
    out = ops.gemm_tn(ad, wd, splits=1)
    tol("vs fp32 math", rel_l2(out, ref), 2e-3)
    nt = ops.gemm(ops.transpose(ad), ops.transpose(wd))
    assert torch.equal(out, nt)
    # strided operands (a column slice of a wider activation), as the backward hands them over
    big_a = torch.zeros(K, M + 64, dtype=torch.bfloat16, device=DEV)
    big_a[:, 32:32 + M] = ad
    big_w = torch.zeros(K, N + 512, dtype=torch.bfloat16, device=DEV)
    big_w[:, 256:256 + N] = wd
    assert torch.equal(ops.gemm_tn(big_a[:, 32:32 + M], big_w[:, 256:256 + N], splits=1), out)
    if K >= 256 and M == 256:
        eye_t = torch.zeros(K, M, dtype=torch.bfloat16)
        eye_t[:M] = torch.eye(M)                                       # A_t^T = [I | 0]: C = the first M rows of W_t
        wa = (torch.arange(K * N).reshape(K, N) % 251 - 125).float().bfloat16()
        assert torch.equal(ops.gemm_tn(eye_t.to(DEV), wa.to(DEV), splits=1).cpu(), wa[:M])
    assert not ops.gemm_tn_ok(ad[:, :M - 8], wd) and not ops.gemm_tn_ok(ad[:K - 8], wd[:K - 8])


@pytest.mark.parametrize("M,N,K", [(256, 256, 64), (256, 512, 128), (768, 256, 1024), (1024, 1536, 2112), (4096, 5120, 7168)])
def test_gemm_nn_weight_read_contraction_major(ops, M, N, K):
    """r04: vita_gemm_bf16_nn: C[M, N] = A[M, K] W[K, N] (the dgrad GEMM, grad_output.matmul(weight), M/core/tensor_parallel/layers.py:444,453,
    with the weight [out, in] as the forward holds it) vs fp32 math, vs the NT kernel fed with a vita_transpose_bf16 copy of W (same MFMA
    order along the contraction: bit-identical), strided operands, an identity case that catches swapped roles, and the dgrad helpers."""
    from long_vita_amd import autograd_fns, training
    a = (torch.randn(M, K, generator=g(400)) * 0.5).bfloat16()
    w = (torch.randn(K, N, generator=g(401)) * (1.0 / math.sqrt(K))).bfloat16()
    ref = (a.float() @ w.float()).bfloat16()
    ad, wd = a.to(DEV), w.to(DEV)
    assert ops.gemm_nn_ok(ad, wd)
    out = ops.gemm_nn(ad, wd)
    tol("vs fp32 math", rel_l2(out, ref), 2e-3)
    nt = ops.gemm(ad, ops.transpose(wd))
    assert torch.equal(out, nt)
    big_a = torch.zeros(M, K + 64, dtype=torch.bfloat16, device=DEV)
    big_a[:, 32:32 + K] = ad
    big_w = torch.zeros(K, N + 512, dtype=torch.bfloat16, device=DEV)
    big_w[:, 256:256 + N] = wd
    assert torch.equal(ops.gemm_nn(big_a[:, 32:32 + K], big_w[:, 256:256 + N]), out)
    if K >= M:
        eye = torch.zeros(M, K, dtype=torch.bfloat16)
        eye[:, :M] = torch.eye(M)                                      # A = [I | 0]: C = the first M rows of W
        wa = (torch.arange(K * N).reshape(K, N) % 251 - 125).float().bfloat16()
        assert torch.equal(ops.gemm_nn(eye.to(DEV), wa.to(DEV)).cpu(), wa[:M])
    # the two dgrad helpers take this path for whole tiles and give what the transposing path gives
    assert torch.equal(autograd_fns.dgrad(ad, wd), nt) and torch.equal(training._dgrad(ad, wd), nt)
    assert not ops.gemm_nn_ok(ad[:M - 8], wd) and not ops.gemm_nn_ok(ad, wd[:, :N - 8])
    ragged = autograd_fns.dgrad(ad[:M - 8].contiguous(), wd)           # rows not a multiple of 256: the transposing fallback
    tol("ragged fallback", rel_l2(ragged, nt[:M - 8]), 2e-3)


@pytest.mark.parametrize("M,N,K,S", [(256, 256, 4096, 4), (512, 256, 8320, 13), (1024, 1024, 66 * 64, 3), (256, 512, 130 * 64, 8)])
def test_gemm_tn_split_k(ops, M, N, K, S):
    """r04: vita_gemm_bf16_tn_splitk — the contraction cut into S ranges (uneven when K / 64 is not a multiple of S), fp32 partials,
    one rounding after the sum — vs fp32 math and vs the one-pass kernel (same products, a different fp32 summation order: within one
    bf16 ulp of each other); the automatic choice (ops.tn_splits) takes the split path for few output tiles over a long contraction."""
    a_t = (torch.randn(K, M, generator=g(300)) * 0.5).bfloat16()
    w_t = (torch.randn(K, N, generator=g(301)) * (1.0 / math.sqrt(K))).bfloat16()
    ref = (a_t.float().t() @ w_t.float())
    ad, wd = a_t.to(DEV), w_t.to(DEV)
    one = ops.gemm_tn(ad, wd, splits=1)
    out = ops.gemm_tn(ad, wd, splits=S)
    tol("split-K vs fp32 math", rel_l2(out, ref), 2e-3)
    tol("split-K vs one pass", rel_l2(out, one), 2.5e-3)
    assert float((out.float() - one.float()).abs().max()) <= float(one.float().abs().max()) * 2 ** -7
    auto = ops.gemm_tn(ad, wd)
    assert ops.tn_splits(M, N, K) > 1
    tol("automatic split vs fp32 math", rel_l2(auto, ref), 2e-3)


def test_colsum_is_the_bias_gradient(ops):
    """r04: vita_colsum_bf16 = grad_output.sum(dim=0) (M/core/tensor_parallel/layers.py:524) in fp32, rows not a multiple of anything,
    a strided view included."""
    x = torch.randn(25933, 1024, generator=g(310)).bfloat16()
    ref = x.float().sum(dim=0)
    out = ops.colsum(x.to(DEV))
    tol("colsum", rel_l2(out, ref), 1e-5)
    big = torch.zeros(4099, 3072 + 64, dtype=torch.bfloat16)
    big[:, 32:32 + 3072] = torch.randn(4099, 3072, generator=g(311)).bfloat16()
    tol("strided colsum", rel_l2(ops.colsum(big.to(DEV)[:, 32:32 + 3072]), big[:, 32:32 + 3072].float().sum(dim=0)), 1e-5)
    assert float(ops.colsum(torch.zeros(0, 256, dtype=torch.bfloat16, device=DEV)).abs().sum()) == 0.0
    # r05 (ABI 18): the ordered form — row-block partials added in block order — gives the same BITS every run (the atomic form did not:
    # the recompute test's bit-identity assertion on linear_qkv.bias failed on it once the GEMM in front of it got faster)
    xd = (torch.randn(70001, 2048, generator=g(312)) * 3).bfloat16().to(DEV)
    first = ops.colsum(xd)
    for _ in range(5):
        assert torch.equal(ops.colsum(xd), first)
    tol("ordered colsum", rel_l2(first, xd.float().sum(dim=0)), 1e-5)
