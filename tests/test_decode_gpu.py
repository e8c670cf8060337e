"""Single-token decode against the sequence-sharded KV cache: kernel parity
vs fp32 torch, and the cached decode loop vs the reference behaviour (re-prefill per token,
M/inference/text_generation/generation.py:127-135) and vs the CPU oracle, at CP = 1 and simulated CP."""
import math

import pytest
import torch

pytestmark = pytest.mark.gpu

from oracle import llm as ollm  # noqa: E402
from test_model_gpu import SMALL, _llm_pair, _run_ranks, rel_l2  # noqa: E402

from conftest import tol  # noqa: E402

DEV = "cuda"


@pytest.fixture(scope="module")
def amd():
    from long_vita_amd import generation, gpt_vl_model, inference_params, ops, parallel_state
    ops._L.load(allow_build=False)
    return dict(ops=ops, gpt=gpt_vl_model, gen=generation, mpu=parallel_state, ip=inference_params)


def bf(t):
    return t.to(torch.bfloat16)


def lp_err(a, b):
    """rel L2 between two log-prob rows after removing the common offset (log-probs sit at about -log V)."""
    a, b = a.float().cpu(), b.float().cpu()
    return rel_l2(a - a.mean(dim=-1, keepdim=True), b - b.mean(dim=-1, keepdim=True))


# ---------------------------------------------------------------------------------------------
@pytest.mark.parametrize("N,K", [(7168, 5120), (1000, 264), (5120, 13824)])
def test_gemv_epilogues(amd, N, K):
    ops = amd["ops"]
    g = torch.Generator().manual_seed(N)
    x = bf(torch.randn(K, generator=g))
    w = bf(torch.randn(N, K, generator=g) * 0.05)
    b = bf(torch.randn(N, generator=g))
    r = bf(torch.randn(N, generator=g))
    acc = w.float() @ x.float()
    xd, wd = x.to(DEV), w.to(DEV)

    def close(out, ref):
        ref = bf(ref).float()
        d = (out.float().cpu() - ref).abs()
        assert float((d / (ref.abs() + 1.0)).max()) < 2e-2          # <= ~1 bf16 ulp of the value
        assert float((d == 0).float().mean()) > 0.9

    close(ops.gemv(xd, wd), acc)
    close(ops.gemv(xd, wd, ops.EPI_BIAS, b.to(DEV)), acc + b.float())
    close(ops.gemv(xd, wd, ops.EPI_RESIDUAL, residual=r.to(DEV)), bf(acc).float() + r.float())
    # in-place residual (out aliases residual), as the decode layer uses it
    rr = r.to(DEV).clone()
    ops.gemv(xd, wd, ops.EPI_RESIDUAL, residual=rr, out=rr)
    close(rr, bf(acc).float() + r.float())
    if N % 2 == 0:
        F = N // 2
        gte, up = bf(acc[:F]).float(), bf(acc[F:]).float()
        close(ops.gemv(xd, wd, ops.EPI_SWIGLU), bf(torch.nn.functional.silu(gte)).float() * up)
    # agrees with the MFMA GEMM on the same row
    if K % 64 == 0:
        big = ops.gemm(xd[None].contiguous(), wd, ops.EPI_BIAS, b.to(DEV))[0]
        tol("ops.gemv(xd, wd, ops.EPI_BIAS, b.to(DEV)), big", rel_l2(ops.gemv(xd, wd, ops.EPI_BIAS, b.to(DEV)), big), 2.3e-04)


def _attn_ref(q, k, v, scale):
    """q [G, qpg, d], k/v [len, G, d] fp32 -> [G*qpg, d]."""
    s = torch.einsum("gqd,lgd->gql", q, k) * scale
    p = torch.softmax(s, dim=-1)
    return torch.einsum("gql,lgd->gqd", p, v).reshape(-1, q.shape[-1])


@pytest.mark.parametrize("length,G,qpg", [(1, 8, 5), (255, 2, 4), (256, 2, 1), (1000, 8, 5), (40000, 8, 5),
                                          (70001, 2, 8), (3000, 4, 7)])
def test_decode_attention_vs_fp32(amd, length, G, qpg):
    ops = amd["ops"]
    d = 128
    g = torch.Generator().manual_seed(length)
    cap = length + 7
    mixed = bf(torch.randn(G, qpg + 2, d, generator=g)).to(DEV)           # q read as a strided view of mixed qkv
    kv = bf(torch.randn(2, cap, G, d, generator=g)).to(DEV)
    q = mixed[:, :qpg]
    pm, pl, po = ops.decode_attn_partial(q, kv[0], kv[1], length)
    assert pm.shape[0] == ops.decode_splits(length)
    ctx = ops.decode_attn_merge(pm, pl, po, True)
    ref = _attn_ref(q.float().cpu(), kv[0, :length].float().cpu(), kv[1, :length].float().cpu(), 1 / math.sqrt(d))
    tol("ctx, ref", rel_l2(ctx, ref), 2.6e-03)


def test_decode_attention_sharded_merge_equals_whole(amd):
    """Two cache shards (a CP = 2 layout, one of them empty on a third 'rank') merged through the packed
    message == attention over the whole cache."""
    ops = amd["ops"]
    G, qpg, d, length = 8, 5, 128, 5000
    g = torch.Generator().manual_seed(3)
    q = bf(torch.randn(G, qpg, d, generator=g)).to(DEV)
    kv = bf(torch.randn(2, length, G, d, generator=g)).to(DEV)
    whole = ops.decode_attn_merge(*ops.decode_attn_partial(q, kv[0], kv[1], length), True)
    cut = 1777
    H = G * qpg
    msgs = torch.empty(3, H * d + 2 * H, dtype=torch.float32, device=DEV)
    ops.decode_attn_merge(*ops.decode_attn_partial(q, kv[0, :cut], kv[1, :cut], cut), False, packed_out=msgs[0])
    ops.decode_attn_merge(*ops.decode_attn_partial(q, kv[0, cut:], kv[1, cut:], length - cut), False,
                          packed_out=msgs[1])
    ops.decode_attn_merge(*ops.decode_attn_partial(q, kv[0], kv[1], 0), False, packed_out=msgs[2])    # empty shard
    gm, gl, go = ops.unpack_partials(msgs, H, d)
    merged = ops.decode_attn_merge(gm, gl, go, True)
    tol("merged, whole", rel_l2(merged, whole), 2e-3)
    assert torch.isfinite(merged.float()).all()


# ---------------------------------------------------------------------------------------------
def _decode_run(amd, model, prompt, n_new, max_len, use_kv_cache, ext=None):
    gen = amd["gen"]
    tokens = torch.zeros(1, max_len, dtype=torch.long, device=DEV)
    P = prompt.shape[1]
    tokens[:, :P] = prompt
    lengths = torch.tensor([P], device=DEV)
    lps = None
    it = gen.generate_tokens_probs_and_return_on_first_stage(model, tokens, lengths, return_output_log_probs=True,
                                                             external_inputs=ext, use_kv_cache=use_kv_cache)
    for i, (toks, _, lps) in enumerate(it):
        if i + 1 == n_new:
            break
    return tokens[:, : P + n_new].clone(), lps[:, P - 1: P - 1 + n_new].clone()


@pytest.mark.parametrize("graph,fused", [(True, False), (False, False), (False, True)])
def test_cached_decode_matches_reprefill_and_oracle(amd, graph, fused):
    """Greedy decode of 6 tokens: the cached loop (token step replayed from a captured HIP graph, or launched
    eagerly) produces the log-probs the reference's re-prefill loop produces on the same tokens, and the oracle's
    full-sequence logits at those positions."""
    cfgd = SMALL
    ocfg, p, model = _llm_pair(amd, cfgd)
    model.decode_graph = graph
    model.decode_fused = fused
    P, n_new, max_len = 300, 6, 512
    prompt = torch.randint(0, cfgd["vocab"], (1, P), generator=torch.Generator().manual_seed(12)).to(DEV)
    toks_c, lp_c = _decode_run(amd, model, prompt, n_new, max_len, True)
    # teacher-forced comparison: re-prefill (reference behaviour) on the cached run's own tokens
    buf = torch.zeros(1, max_len, dtype=torch.long, device=DEV)
    buf[:, : P + n_new] = toks_c
    for j in range(n_new):
        ref_logits = amd["gen"].prefill_step(model, buf, P + j, None, reference_compat=False)
        ref_lp = torch.log_softmax(ref_logits.float(), dim=1)
        assert lp_err(lp_c[:, j], ref_lp) < 1.5e-2, (j, lp_err(lp_c[:, j], ref_lp))
    ora = ollm.prefill_logits(toks_c.cpu(), p, ocfg, list(range(P - 1, P - 1 + n_new)))[0]
    ora_lp = torch.log_softmax(ora.float(), dim=1)
    assert lp_err(lp_c[0], ora_lp) < 2.5e-2
    # greedy tokens are the argmax of the oracle's logits wherever the oracle's top-2 margin is not a near-tie
    top2 = ora.float().topk(2, dim=1).values
    clear = (top2[:, 0] - top2[:, 1]) > 0.05 * ora.float().std()
    assert bool((toks_c[0, P:].cpu()[clear] == ora.argmax(dim=1)[clear]).all())


def test_cached_decode_with_visual_prompt(amd):
    """external_inputs are consumed by the prefill only (gpt_vl_model.py:262: `not key_value_memory_dict`)."""
    from long_vita_amd import synthetic, vision
    cfgd = SMALL
    ocfg, p, model = _llm_pair(amd, cfgd)
    vit = vision.MegatronVisionModel.random_init(vision.VisionConfig(num_layers=1, llm_hidden=cfgd["hidden"]), seed=4,
                                                 device=DEV)
    model.external_feature_model = vit
    S, n_new, max_len = 640, 3, 1024
    prompt, ext = synthetic.make_request(S, 2, seed=2, device=DEV)
    prompt = prompt % cfgd["vocab"]
    toks_c, lp_c = _decode_run(amd, model, prompt, n_new, max_len, True, ext)
    toks_r, lp_r = _decode_run(amd, model, prompt, n_new, max_len, False, ext)
    assert lp_err(lp_c[:, 0], lp_r[:, 0]) < 1.5e-2
    if torch.equal(toks_c, toks_r):                        # same greedy path -> every step comparable
        assert lp_err(lp_c, lp_r) < 1.5e-2


@pytest.mark.parametrize("cp,P,fused", [(2, 1500, True), (2, 1024, False), (4, 2300, True), (8, 4500, True)])
def test_cached_decode_context_parallel(amd, monkeypatch, cp, P, fused):
    """Simulated CP ranks: padded zig-zag prefill fills the cache shards (pad rows dropped), generated tokens
    are appended round-robin, per-rank partials merged after the all-gather == CP = 1 cached decode."""
    cfgd = SMALL
    ocfg, p, model1 = _llm_pair(amd, cfgd)
    G = amd["gpt"]
    n_new, max_len = 5, 8192 if cp == 8 else 4096
    prompt = torch.randint(0, cfgd["vocab"], (1, P), generator=torch.Generator().manual_seed(P)).to(DEV)
    toks1, lp1 = _decode_run(amd, model1, prompt, n_new, max_len, True)

    def rank_fn(r):
        m = G.GPTVLModel(model1.cfg, model1.p)
        m.decode_fused = fused
        return _decode_run(amd, m, prompt, n_new, max_len, True)

    outs = _run_ranks(cp, rank_fn, amd, monkeypatch)
    for r in range(1, cp):
        assert torch.equal(outs[r][0], outs[0][0]) and torch.equal(outs[r][1], outs[0][1])
    assert lp_err(outs[0][1][:, 0], lp1[:, 0]) < 1.5e-2
    if torch.equal(outs[0][0], toks1):
        assert lp_err(outs[0][1], lp1) < 1.5e-2
    ora = ollm.prefill_logits(outs[0][0].cpu(), p, ocfg, list(range(P - 1, P - 1 + n_new)))[0]
    assert lp_err(outs[0][1][0], torch.log_softmax(ora.float(), dim=1)) < 2.5e-2


def test_fused_decode_layer_equals_separate_kernels(amd):
    """vita_decode_layer_attn / _mlp run the same kernels with the same rounding chains: logits of the fused and the
    kernel-by-kernel token step agree to the last bf16 bit for almost every element."""
    cfgd = SMALL
    ocfg, p, model = _llm_pair(amd, cfgd)
    P, n_new, max_len = 700, 4, 1024
    prompt = torch.randint(0, cfgd["vocab"], (1, P), generator=torch.Generator().manual_seed(3)).to(DEV)
    model.decode_fused = True
    toks_f, lp_f = _decode_run(amd, model, prompt, n_new, max_len, True)
    model.decode_fused = False
    toks_s, lp_s = _decode_run(amd, model, prompt, n_new, max_len, True)
    assert torch.equal(toks_f, toks_s)
    assert lp_err(lp_f, lp_s) < 2e-3


@pytest.mark.parametrize("tp,cp", [(2, 1), (2, 2)])
def test_cached_decode_tensor_parallel(amd, monkeypatch, tp, cp):
    """TP in the decode path (the released server runs TP 8 x CP 4, R/scripts/megatron/qwen25/..._server_cp.sh:102-104): every
    tensor-parallel rank keeps the K/V rows of ITS kv groups, the row-parallel GEMVs all-reduce their bf16 partial sums before the
    residual add, the vocab-parallel logits are gathered — simulated ranks == TP = 1 cached decode (and the oracle's logits)."""
    from long_vita_amd import tensor_parallel as tpar
    from test_train_gpu import _run_grid
    cfgd = dict(SMALL, kv_groups=4)                        # 4 kv groups: 2 per tensor-parallel rank
    ocfg, p, model1 = _llm_pair(amd, cfgd)
    G = amd["gpt"]
    P, n_new, max_len = 1536, 5, 4096
    prompt = torch.randint(0, cfgd["vocab"], (1, P), generator=torch.Generator().manual_seed(P)).to(DEV)
    toks1, lp1 = _decode_run(amd, model1, prompt, n_new, max_len, True)

    def rank_fn(ci, ti):
        shard, cfg_l = tpar.shard_llm_params(p, G.GPTConfig(**cfgd), tp, ti)
        m = G.GPTVLModel.from_oracle_layout(cfg_l, shard, None, DEV)
        return _decode_run(amd, m, prompt, n_new, max_len, True)

    outs = _run_grid(tp, cp, rank_fn, amd, monkeypatch)
    first = outs[(0, 0)]
    for key, o in outs.items():
        assert torch.equal(o[0], first[0]) and torch.equal(o[1], first[1]), key
    assert lp_err(first[1][:, 0], lp1[:, 0]) < 1.5e-2
    if torch.equal(first[0], toks1):
        assert lp_err(first[1], lp1) < 1.5e-2
    ora = ollm.prefill_logits(first[0].cpu(), p, ocfg, list(range(P - 1, P - 1 + n_new)))[0]
    assert lp_err(first[1][0], torch.log_softmax(ora.float(), dim=1)) < 2.5e-2
