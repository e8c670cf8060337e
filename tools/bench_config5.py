"""BASELINE config 5 (Long-VITA-128K training step, TP = 2 x CP = 4) — what ONE MI355X can show of it:

  rank      the per-rank pieces of one decoder layer at config 5's geometry, timed with HIP events:
              * attention forward + backward: S_l = 32768 local queries (zig-zag chunks r, 7 - r of 16384) against the 131072
                gathered keys through the CP = 4 chunk tables, 20 query : 4 kv heads (TP = 2 halves 40 : 8), dK / dV in the
                gathered layout (what the reduce-scatter consumes);
              * the tensor-parallel GEMMs at M = 32768 rows: forward, dgrad and wgrad shapes of qkv / proj / fc1 / fc2;
              * the HBM-bound rest (RMSNorm fwd / bwd, RoPE, SwiGLU bwd) at 32768 rows;
            -> a modelled per-rank step time (forward + recompute of `--recompute-num-layers` + backward), written next to the raw
            numbers.  Communication (K/V all-gather + dK/dV reduce-scatter over 4 ranks, TP all-reduce / reduce-scatter) is NOT in
            it: one GPU cannot measure it.
  step      one whole 128K training step on one GPU (TP = CP = 1, 48 layers, 512 answer tokens, logits-masked head,
            `--recompute-granularity full --recompute-method block --recompute-num-layers N`), wall-clock.

  rankstep  (r05) the WHOLE training step of ONE emulated rank (tp 0 of 2, cp 1 of 4): embedding, 48 TP-halved layers at S_l = 32768 with the
            CP = 4 chunk tables against 131072 gathered keys, `--recompute-num-layers N`, final norm, vocab-parallel head, loss — every
            kernel the real rank would run, with LOCAL stand-ins for the collectives (all-gather = this rank's message copied into every
            slot, reduce-scatter = this rank's slice, all-reduce = identity: the bytes move inside the GPU, nothing waits for a peer), so
            the figure is the rank's compute time with zero exposed communication — to be put beside the modelled one.

    python tools/bench_config5.py rank [step [N]] [rankstep [N]]
Writes JSON lines to gpurun_out/r06_config5.jsonl."""
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402

from long_vita_amd import gpt_vl_model, lib, ops, training  # noqa: E402

DEV = "cuda:0"
OUT = os.path.join(ROOT, "gpurun_out")
os.makedirs(OUT, exist_ok=True)
LOG = open(os.path.join(OUT, "r06_config5.jsonl"), "a")
lib.load(allow_build=False)


def emit(**kw):
    s = json.dumps(kw)
    print(s, flush=True)
    LOG.write(s + "\n")
    LOG.flush()


def timeit(fn, warmup=1, iters=3):
    for _ in range(warmup):
        fn()
    torch.cuda.synchronize()
    ts = []
    for _ in range(iters):
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record(); fn(); b.record()
        torch.cuda.synchronize()
        ts.append(a.elapsed_time(b))
    return sorted(ts)[len(ts) // 2]


def rnd(*shape, seed=0, scale=1.0):
    g = torch.Generator(device=DEV).manual_seed(seed)
    return (torch.randn(*shape, generator=g, device=DEV) * scale).bfloat16()


def bench_rank(rank=1):
    cp, tp, S, D = 4, 2, 131072, 128
    hq, hkv, hidden, ffn = 40 // tp, 8 // tp, 5120, 13824 // tp
    c = S // (2 * cp)
    s_l = 2 * c
    q = rnd(1, s_l, hq, D, seed=1)
    rows = rnd(cp * 2 * s_l, hkv, D, seed=2)                    # gathered [rank p][K | V][S_l][4][128]
    d_o = rnd(1, s_l, hq, D, seed=3)
    kv_gid, kv_row = [], []
    for p in range(cp):
        kv_gid += [p, 2 * cp - 1 - p]
        kv_row += [p * 2 * s_l, p * 2 * s_l + c]
    geo = dict(chunk_len=c, q_chunk_gid=[rank, 2 * cp - 1 - rank], kv_chunk_gid=kv_gid, kv_chunk_row=kv_row)
    k_all, v_all = rows.unsqueeze(0), rows[s_l:].unsqueeze(0)
    o = torch.empty_like(q)
    lse = torch.empty(1, hq, s_l, dtype=torch.float32, device=DEV)
    pairs = c * (rank + 0.5) * c + c * (2 * cp - 1 - rank + 0.5) * c
    unit = 2.0 * D * hq * pairs
    t_f = timeit(lambda: ops.flash_attn(q, k_all, v_all, causal=True, out=o, lse_out=lse, **geo))
    emit(kind="cfg5_attn_fwd", rank=rank, s_local=s_l, keys=S, heads=f"{hq}:{hkv}", ms=t_f, algorithmic_tflops=2 * unit / t_f / 1e9)
    dq, d_rows = torch.empty_like(q), torch.empty_like(rows)
    t_b = timeit(lambda: ops.flash_attn_bwd(q, k_all, v_all, o, d_o, lse, dq5=dq, dk=d_rows.unsqueeze(0), dv=d_rows[s_l:].unsqueeze(0), **geo))
    emit(kind="cfg5_attn_bwd", rank=rank, s_local=s_l, keys=S, heads=f"{hq}:{hkv}", ms=t_b, algorithmic_tflops=5 * unit / t_b / 1e9,
         note="delta pre-pass + dQ kernel + dK kernel + dV kernel; dK / dV written for all 131072 gathered rows")
    del q, rows, d_o, o, dq, d_rows
    # ---- tensor-parallel GEMMs of one layer at M = S_l rows (sequence parallelism gathers the TP shard before the column-parallel GEMMs)
    M = s_l
    qkv_n = (hq + 2 * hkv) * D
    shapes = {"qkv": (qkv_n, hidden, ops.EPI_BIAS), "proj": (hidden, hq * D, ops.EPI_NONE), "fc1": (2 * ffn, hidden, ops.EPI_NONE),
              "fc2": (hidden, ffn, ops.EPI_NONE)}
    tot = {"fwd": 0.0, "dgrad": 0.0, "wgrad": 0.0}
    for name, (N, K, epi) in shapes.items():
        a, w = rnd(M, K, seed=5, scale=0.5), rnd(N, K, seed=6, scale=0.02)
        bias = rnd(N, seed=7) if epi == ops.EPI_BIAS else None
        out = torch.empty(M, N, dtype=torch.bfloat16, device=DEV)
        t = timeit(lambda: ops.gemm(a, w, epi, bias, out=out))
        dy = rnd(M, N, seed=8)
        wt = ops.transpose(w)                                   # dgrad: dy [M, N] @ w [N, K] = gemm(dy, w^T [K, N])
        dx = torch.empty(M, K, dtype=torch.bfloat16, device=DEV)
        t_d = timeit(lambda: ops.gemm(dy, ops.transpose(w), out=dx))
        t_w = timeit(lambda: training._wgrad_tn(dy, a))                          # wgrad: dy^T [N, M] @ a [M, K] (vita_gemm_bf16_tn, r03)
        fl = 2.0 * M * N * K
        emit(kind="cfg5_gemm", name=name, M=M, N=N, K=K, fwd_ms=t, dgrad_ms=t_d, wgrad_ms=t_w, fwd_tflops=fl / t / 1e9,
             dgrad_tflops=fl / t_d / 1e9, wgrad_tflops=fl / t_w / 1e9, note="dgrad includes the weight transpose it needs; wgrad = vita_gemm_bf16_tn on the operands as they are")
        tot["fwd"] += t; tot["dgrad"] += t_d; tot["wgrad"] += t_w
        del a, w, out, dy, wt, dx
    # ---- HBM-bound kernels at M rows (norms run on the rank's S_l / TP rows under sequence parallelism: count M / TP rows)
    x = rnd(M // tp, hidden, seed=9)
    wn = torch.ones(hidden, dtype=torch.bfloat16, device=DEV)
    t_n = timeit(lambda: ops.rmsnorm(x, wn, 1e-6))
    dwn = torch.zeros(hidden, dtype=torch.float32, device=DEV)
    t_nb = timeit(lambda: ops.rmsnorm_bwd(x, x, wn, 1e-6, dwn))
    y = rnd(M, 2 * ffn, seed=10)
    da = rnd(M, ffn, seed=11)
    t_s, t_sb = timeit(lambda: ops.swiglu(y)), timeit(lambda: ops.swiglu_bwd(y, da))
    mixed = rnd(M, qkv_n, seed=12)
    cos, sin = ops.rope_table(torch.arange(M, device=DEV), ops.rope_inv_freq(D, 1e6, DEV))
    t_r = timeit(lambda: ops.rope_qkv_(mixed, hkv, hq // hkv, D, cos, sin, None, 1))
    emit(kind="cfg5_hbm", rmsnorm_fwd_ms=t_n, rmsnorm_bwd_ms=t_nb, swiglu_fwd_ms=t_s, swiglu_bwd_ms=t_sb, rope_qkv_ms=t_r)
    small_f = 2 * t_n + t_r + t_s
    small_b = 2 * t_nb + t_r + t_sb
    fwd = t_f + tot["fwd"] + small_f
    bwd = t_b + tot["dgrad"] + tot["wgrad"] + small_b
    for n_rec in (20, 48):
        step = 48 * (fwd + bwd) + n_rec * fwd
        emit(kind="cfg5_model", recompute_num_layers=n_rec, layer_fwd_ms=fwd, layer_bwd_ms=bwd, modelled_rank_step_s=step / 1e3,
             tokens_per_s_node=131072 / (step / 1e3),
             note="48 x (forward + backward) + N x recompute forward of the per-rank kernels measured above; communication, embedding, "
                  "head and loss not included")


def bench_step(n_rec):
    cfg = gpt_vl_model.GPTConfig()
    model = gpt_vl_model.GPTVLModel.random_init(cfg, seed=1, device=DEV)
    g = torch.Generator(device=DEV).manual_seed(2)
    S, ans = 131072, 512

    def batch(s):
        tokens = torch.randint(0, 151643, (1, s), generator=g, device=DEV)
        labels = torch.roll(tokens, -1, 1)
        loss_mask = torch.zeros(1, s, device=DEV)
        loss_mask[0, s - min(ans, s // 2):] = 1
        return tokens, labels, loss_mask

    step = training.TrainStep(model, recompute_num_layers=n_rec)
    loss, grads = step.forward_backward(*batch(2048))          # warm-up at 2K: loads every kernel, sizes nothing for 128K
    del grads
    model._ws = {}
    torch.cuda.synchronize()
    torch.cuda.reset_peak_memory_stats()
    b = batch(S)
    t0 = time.perf_counter()
    loss, grads = step.forward_backward(*b)
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    del grads
    lin = cfg.num_layers * 2 * (cfg.hidden * cfg.qkv_out + cfg.hidden * cfg.heads * cfg.head_dim + 3 * cfg.hidden * cfg.ffn) * S
    attn = cfg.num_layers * 4 * cfg.head_dim * cfg.heads * (S * (S + 1) // 2)
    r = n_rec / cfg.num_layers
    alg = (3 + r) * lin + (3.5 + r) * attn
    emit(kind="train_step_128k", what=f"fwd + bwd, 48 layers, TP = CP = 1, recompute block = {n_rec} layers, one timed step", seq=S,
         answer_tokens=ans, recompute_num_layers=n_rec, s_per_step=dt, loss=float(loss), algorithmic_pflop_per_step=alg / 1e15,
         tflops=alg / dt / 1e12, tflops_without_recompute_work=(3 * lin + 3.5 * attn) / dt / 1e12, tokens_per_s=S / dt,
         peak_mem_gb=torch.cuda.max_memory_allocated() / 2 ** 30)


def bench_rankstep(n_rec, tp=2, cp=4, tp_rank=0, cp_rank=1):
    import types
    import torch.distributed as dist
    from long_vita_amd import parallel_state as mpu, tensor_parallel as tpar

    class Work:
        def wait(self):
            return True

    def group(n, r):
        return types.SimpleNamespace(size=n, rank=r)

    def all_gather_into_tensor(out, inp, group=None, async_op=False):
        flat = out.view(group.size, -1)
        for q in range(group.size):
            flat[q].copy_(inp.reshape(-1))
        return Work() if async_op else None

    def reduce_scatter_tensor(out, inp, group=None, async_op=False, op=None):
        out.view(-1).copy_(inp.view(group.size, -1)[group.rank])
        return Work() if async_op else None

    def all_reduce(t, group=None, op=None, async_op=False):
        return Work() if async_op else None

    dist.all_gather_into_tensor, dist.reduce_scatter_tensor, dist.all_reduce = all_gather_into_tensor, reduce_scatter_tensor, all_reduce
    full_cfg = gpt_vl_model.GPTConfig()
    full = gpt_vl_model.GPTVLModel.random_init(full_cfg, seed=1, device=DEV)
    shard, cfg_l = tpar.shard_llm_params(full.p, full_cfg, tp, tp_rank)
    del full
    torch.cuda.empty_cache()
    model = gpt_vl_model.GPTVLModel(cfg_l, shard)
    mpu.set_context_parallel_state(cp, cp_rank, group(cp, cp_rank))
    mpu.set_tensor_parallel_state(tp, tp_rank, group(tp, tp_rank))
    g = torch.Generator(device=DEV).manual_seed(2)
    S, ans = 131072, 512
    tokens = torch.randint(0, 151643, (1, S), generator=g, device=DEV)
    labels = torch.roll(tokens, -1, 1)
    loss_mask = torch.zeros(1, S, device=DEV)
    loss_mask[0, S - ans:] = 1
    step = training.TrainStep(model, recompute_num_layers=n_rec)
    times = []
    for i in range(3):
        torch.cuda.synchronize()
        if i == 1:
            torch.cuda.reset_peak_memory_stats()
        t0 = time.perf_counter()
        loss, grads = step.forward_backward(tokens, labels, loss_mask)
        torch.cuda.synchronize()
        times.append(time.perf_counter() - t0)
        del grads
    emit(kind="cfg5_rank_step", what=f"the whole training step of one emulated rank (tp {tp_rank} of {tp}, cp {cp_rank} of {cp}): 48 layers, S = {S} "
         f"(S_l = {S // cp}), recompute block = {n_rec} layers, collectives replaced by local copies (no peer, nothing exposed)",
         recompute_num_layers=n_rec, s_per_step=min(times[1:]), s_per_step_all=times, loss=float(loss),
         tokens_per_s_node_at_zero_exposed_comm=S / min(times[1:]), peak_mem_gb=torch.cuda.max_memory_allocated() / 2 ** 30)
    mpu.destroy_model_parallel()


if __name__ == "__main__":
    which = sys.argv[1:] or ["rank"]
    if "rankstep" in which:
        i = which.index("rankstep")
        n = int(which[i + 1]) if len(which) > i + 1 and which[i + 1].isdigit() else 20
        r = int(which[i + 2]) if len(which) > i + 2 and which[i + 2].isdigit() else 1     # cp rank: 0 holds the answer tokens (the last chunk)
        bench_rankstep(n, cp_rank=r)
    if "rank" in which:
        bench_rank()
    if "step" in which:
        i = which.index("step")
        bench_step(int(which[i + 1]) if len(which) > i + 1 else 48)
