"""Kernel micro-benchmarks on the GPU box (HIP events on the launch stream): GEMM shapes of the 14B
decoder / ViT, flash attention at the BASELINE sequence lengths, HBM-bound kernels.
Writes one JSON line per case to stdout (and gpurun_out/microbench.jsonl)."""
import json
import math
import os
os.environ.setdefault("VITA_DEBUG", "1")      # developer switches (VITA_GEMM_*, VITA_ATTN_*) are honoured only with this set
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402

from long_vita_amd import ops  # noqa: E402

DEV = "cuda"
OUT = os.path.join(ROOT, "gpurun_out")
os.makedirs(OUT, exist_ok=True)
LOG = open(os.path.join(OUT, "microbench.jsonl"), "a")


def timeit(fn, warmup=2, iters=5):
    for _ in range(warmup):
        fn()
    torch.cuda.synchronize()
    evs = []
    for _ in range(iters):
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record(); fn(); b.record()
        evs.append((a, b))
    torch.cuda.synchronize()
    ts = sorted(a.elapsed_time(b) for a, b in evs)
    return ts[len(ts) // 2], ts[0]


def emit(**kw):
    s = json.dumps(kw)
    print(s, flush=True)
    LOG.write(s + "\n"); LOG.flush()


def bench_gemm(M, N, K, epi=0, tag=""):
    a = (torch.randn(M, K, device=DEV) * 0.5).bfloat16()
    w = (torch.randn((2 * N if epi == 5 else N), K, device=DEV) * 0.02).bfloat16()
    out = torch.empty(M, N, dtype=torch.bfloat16, device=DEV)
    res = torch.randn(M, N, device=DEV).bfloat16() if epi == 3 else None
    bias = torch.randn(N, device=DEV).bfloat16() if epi == 1 else None
    med, best = timeit(lambda: ops.gemm(a, w, epi, bias, None, res, out=out))
    fl = 2.0 * M * (2 * N if epi == 5 else N) * K
    emit(kind="gemm", tag=tag, M=M, N=N, K=K, epi=epi, ms=med, ms_best=best, tflops=fl / med / 1e9)


def bench_attn(S, Hq=40, Hkv=8, D=128, causal=True, B=1, tag=""):
    q = torch.randn(B, S, Hq, D, device=DEV).bfloat16()
    k = torch.randn(B, S, Hkv, D, device=DEV).bfloat16()
    v = torch.randn(B, S, Hkv, D, device=DEV).bfloat16()
    o = torch.empty_like(q)
    med, best = timeit(lambda: ops.flash_attn(q, k, v, causal=causal, out=o), warmup=1, iters=3)
    pairs = S * (S + 1) / 2 if causal else S * S
    fl = 4.0 * D * Hq * pairs * B
    emit(kind="attn", tag=tag, S=S, Hq=Hq, Hkv=Hkv, D=D, causal=causal, B=B, ms=med, ms_best=best, tflops=fl / med / 1e9)


def bench_attn_bwd(S, Hq=40, Hkv=8, D=128, tag=""):
    """Attention backward (dQ pass + dK / dV pass + the delta pre-pass) on a plain causal sequence; VITA_ATTN_BWD_ONLY=dq|dkv
    times one pass alone (developer switch in attn_bwd.hip).  flops: 5 GEMM units of the forward's 2 (algorithmic); the 64-rows-per-wave
    kernels executed 8 through r03 (dQ: S, dP, dQ; dK: S, dP, dK; dV: S, dV) and execute 7 since r04 (attn_bwd_kvp.hip: dK + dV in one
    launch, S once; VITA_ATTN_BWD_KVP=0 restores the two launches), the general kernels (VITA_ATTN_BWD64=0) 7."""
    q = torch.randn(1, S, Hq, D, device=DEV).bfloat16()
    k = torch.randn(1, S, Hkv, D, device=DEV).bfloat16()
    v = torch.randn(1, S, Hkv, D, device=DEV).bfloat16()
    o, lse = ops.flash_attn(q, k, v, causal=True, return_lse=True)
    d_o = torch.randn_like(o)
    dq, dk, dv = torch.empty_like(q), torch.empty_like(k), torch.empty_like(v)
    pairs = S * (S + 1) / 2
    unit = 2.0 * D * Hq * pairs
    fast = os.environ.get("VITA_ATTN_BWD64", "1") != "0"
    two_launches = fast and os.environ.get("VITA_ATTN_BWD_KVP", "1") == "0"
    for only, units in (("", 8 if two_launches else 7), ("dq", 3), ("dkv", 5 if two_launches else 4)):
        if only:
            os.environ["VITA_ATTN_BWD_ONLY"] = only
        med, best = timeit(lambda: ops.flash_attn_bwd(q, k, v, o, d_o, lse, dq5=dq, dk=dk, dv=dv), warmup=1, iters=3)
        os.environ.pop("VITA_ATTN_BWD_ONLY", None)
        emit(kind="attn_bwd", tag=tag, only=only or "both", S=S, ms=med, ms_best=best, executed_tflops=units * unit / med / 1e9,
             algorithmic_tflops=(5 * unit / med / 1e9) if not only else None)


def bench_gemm_tn():
    """wgrad GEMMs: vita_gemm_bf16_tn on the operands as the forward left them vs two vita_transpose_bf16 passes + the NT GEMM (r02's path),
    at the 16K single-GPU shapes and BASELINE config 5's per-rank shapes (tokens = 32768, TP = 2 widths)."""
    for (T, N, K, tag) in [(16384, 7168, 5120, "S16K/qkv"), (16384, 5120, 5120, "S16K/proj"), (16384, 27648, 5120, "S16K/fc1"),
                           (16384, 5120, 13824, "S16K/fc2"), (32768, 3584, 5120, "cfg5/qkv"), (32768, 5120, 2560, "cfg5/proj"),
                           (32768, 13824, 5120, "cfg5/fc1"), (32768, 5120, 6912, "cfg5/fc2"), (131072, 7168, 5120, "S128K/qkv")]:
        dy = torch.randn(T, N, device=DEV).bfloat16()
        x = (torch.randn(T, K, device=DEV) * 0.5).bfloat16()
        out = torch.empty(N, K, dtype=torch.bfloat16, device=DEV)
        med, best = timeit(lambda: ops.gemm_tn(dy, x, out=out))
        ref = ops.gemm(ops.transpose(dy), ops.transpose(x))
        same = bool(torch.equal(ref, out))
        med2, best2 = timeit(lambda: ops.gemm(ops.transpose(dy), ops.transpose(x)))
        med3, _ = timeit(lambda: (ops.transpose(dy), ops.transpose(x)))
        fl = 2.0 * T * N * K
        emit(kind="gemm_tn", tag=tag, tokens=T, N=N, K=K, tn_ms=med, tn_tflops=fl / med / 1e9, transposes_plus_nt_ms=med2,
             transposes_ms=med3, nt_alone_tflops=fl / max(med2 - med3, 1e-6) / 1e9, speedup=med2 / med, bit_identical=same)
        del dy, x, out, ref


def bench_hbm():
    rows, cols = 131072, 5120
    x = torch.randn(rows, cols, device=DEV).bfloat16()
    w = torch.ones(cols, device=DEV).bfloat16()
    y = torch.empty_like(x)
    med, _ = timeit(lambda: ops.rmsnorm(x, w, 1e-6, out=y))
    emit(kind="rmsnorm", rows=rows, cols=cols, ms=med, gbps=2 * rows * cols * 2 / med / 1e6)
    qkv = torch.randn(rows, 7168, device=DEV).bfloat16()
    inv = ops.rope_inv_freq(128, 1e6, DEV)
    cos, sin = ops.rope_table(torch.arange(rows, device=DEV), inv)
    kv = torch.empty(2, rows, 8, 128, dtype=torch.bfloat16, device=DEV)
    med, _ = timeit(lambda: ops.rope_qkv_(qkv, 8, 5, 128, cos, sin, kv))
    emit(kind="rope_qkv", rows=rows, ms=med, gbps=(rows * 6144 * 2 * 2 + rows * 2048 * 2 * 2) / med / 1e6)
    idx = torch.randint(0, 152064, (rows,), device=DEV)
    table = torch.randn(152064, cols, device=DEV).bfloat16()
    med, _ = timeit(lambda: ops.row_gather(table, idx, out=y, check_bounds=False))
    emit(kind="row_gather", rows=rows, cols=cols, ms=med, gbps=2 * rows * cols * 2 / med / 1e6)
    a = torch.randn(2, cols, device=DEV).bfloat16()
    med, _ = timeit(lambda: ops.gemm_skinny(a, table))
    emit(kind="gemm_skinny", M=2, N=152064, K=cols, ms=med, gbps=152064 * cols * 2 / med / 1e6)


def bench_peaks():
    """'achievable peak' references measured on the same box — the vendor GEMM library (hipBLASLt /
    rocBLAS behind torch.matmul, bf16 in, fp32 accumulate) on the decoder's shapes and on a square problem, this
    library's kernel beside it, and a device-to-device copy / a read-only reduction for the HBM ceiling.
    torch.matmul is a yardstick here, not part of the product path."""
    for (M, N, K, tag) in [(8192, 8192, 8192, "square8k"), (131072, 5120, 5120, "S128K/o"), (131072, 27648, 5120, "S128K/fc1"),
                           (131072, 5120, 13824, "S128K/fc2"), (16384, 7168, 5120, "S16K/qkv")]:
        a = (torch.randn(M, K, device=DEV) * 0.5).bfloat16()
        w = (torch.randn(N, K, device=DEV) * 0.02).bfloat16()
        out = torch.empty(M, N, dtype=torch.bfloat16, device=DEV)
        med, best = timeit(lambda: torch.matmul(a, w.t(), out=out))
        emit(kind="peak_gemm_vendor", tag=tag, M=M, N=N, K=K, ms=med, ms_best=best, tflops=2.0 * M * N * K / med / 1e9)
        med2, best2 = timeit(lambda: ops.gemm(a, w, 0, None, None, None, out=out))
        emit(kind="peak_gemm_ours", tag=tag, M=M, N=N, K=K, ms=med2, ms_best=best2, tflops=2.0 * M * N * K / med2 / 1e9,
             ours_over_vendor=med / med2)
        del a, w, out
    n = 1 << 31                                                       # 2 GiB each way: far beyond the 256 MB MALL
    src = torch.empty(n, dtype=torch.uint8, device=DEV).random_(0, 255)
    dst = torch.empty_like(src)
    med, best = timeit(lambda: dst.copy_(src))
    emit(kind="peak_hbm_copy", bytes_each_way=n, ms=med, ms_best=best, gbps_read_plus_write=2 * n / med / 1e6)
    rows = (n // 2) // 5120
    x = src.view(torch.bfloat16)[:rows * 5120].view(rows, 5120)
    y = dst.view(torch.bfloat16)[:rows * 5120].view(rows, 5120)
    w = torch.ones(5120, device=DEV).bfloat16()
    med, best = timeit(lambda: ops.rmsnorm(x, w, 1e-6, out=y))
    emit(kind="peak_hbm_rmsnorm_ours", bytes_each_way=rows * 5120 * 2, ms=med, ms_best=best, gbps_read_plus_write=2 * rows * 5120 * 2 / med / 1e6)
    f = src.view(torch.float32)
    med, best = timeit(lambda: f.sum())
    emit(kind="peak_hbm_read_only", bytes=n, ms=med, ms_best=best, gbps=n / med / 1e6)


def bench_gemm_variants():
    """Experiment: placement of the LDS-DMA pieces in the 256x256 GEMM (VITA_GEMM_EXP, see gemm.hip)."""
    for (M, N, K, tag) in [(131072, 5120, 5120, "S128K/o"), (131072, 5120, 13824, "S128K/fc2"), (16384, 7168, 5120, "S16K/qkv"),
                           (8192, 8192, 8192, "square8k")]:
        a = (torch.randn(M, K, device=DEV) * 0.5).bfloat16()
        w = (torch.randn(N, K, device=DEV) * 0.02).bfloat16()
        out = torch.empty(M, N, dtype=torch.bfloat16, device=DEV)
        ref = torch.matmul(a[:4096], w.t()).float()
        for v in os.environ.get("VARIANTS", "0,4").split(","):
            os.environ["VITA_GEMM_EXP"] = v
            out.zero_()
            med, best = timeit(lambda: ops.gemm(a, w, 0, None, None, None, out=out))
            err = float((out[:4096].float() - ref).abs().max())
            tail = float((out[-256:].float() - torch.matmul(a[-256:], w.t()).float()).abs().max())
            emit(kind="gemm_dma_variant", variant=v, tag=tag, ms=med, ms_best=best, tflops=2.0 * M * N * K / med / 1e9,
                 max_abs_err_vs_vendor=err, tail_err=tail)
        os.environ.pop("VITA_GEMM_EXP")
        del a, w, out


LLM_SHAPES = [(131072, 7168, 5120, 1, "S128K/qkv"), (131072, 5120, 5120, 3, "S128K/o"),
              (131072, 13824, 5120, 5, "S128K/fc1_swiglu"), (131072, 5120, 13824, 3, "S128K/fc2"),
              (16384, 7168, 5120, 1, "S16K/qkv"), (16384, 5120, 13824, 3, "S16K/fc2")]
VIT_SHAPES = [(262400, 3072, 1024, 1, "vit256f/qkv"), (262400, 1024, 1024, 4, "vit256f/proj"), (262400, 4096, 1024, 2, "vit256f/fc1_gelu"),
              (262400, 1024, 4096, 4, "vit256f/fc2"), (65536, 1024, 4096, 0, "projector/fc1"), (65536, 5120, 1024, 0, "projector/fc2")]


def bench_gemm_env(name, values, shapes=None):
    """Sweep one developer env switch of the GEMM (VITA_GEMM_STAGGER, VITA_GEMM_KERNEL, ...) over GEMMs with their epilogues."""
    for (M, N, K, epi, tag) in (shapes or LLM_SHAPES):
        a = (torch.randn(M, K, device=DEV) * 0.5).bfloat16()
        w = (torch.randn((2 * N if epi == 5 else N), K, device=DEV) * 0.02).bfloat16()
        out = torch.empty(M, N, dtype=torch.bfloat16, device=DEV)
        res = torch.randn(M, N, device=DEV).bfloat16() if epi in (3, 4) else None
        bias = torch.randn(N, device=DEV).bfloat16() if epi in (1, 2, 4) else None
        scale = torch.randn(N, device=DEV).bfloat16() if epi == 4 else None
        base = None
        for v in values:
            os.environ[name] = v
            out.zero_()
            med, best = timeit(lambda: ops.gemm(a, w, epi, bias, scale, res, out=out))
            if base is None:
                base = out[:2048].float().clone()
            err = float((out[:2048].float() - base).abs().max())
            fl = 2.0 * M * (2 * N if epi == 5 else N) * K
            emit(kind="gemm_env", env=name, value=v, tag=tag, ms=med, ms_best=best, tflops=fl / med / 1e9, max_abs_diff_vs_first=err)
        os.environ.pop(name)
        del a, w, out


if __name__ == "__main__":
    which = sys.argv[1:] or ["gemm", "attn", "hbm"]
    if which and which[0] in ("env", "envvit"):
        bench_gemm_env(which[1], which[2].split(","), VIT_SHAPES if which[0] == "envvit" else None)
        sys.exit(0)
    if "variants" in which:
        bench_gemm_variants()
    if "peaks" in which:
        bench_peaks()
    if "gemm_tn" in which:
        bench_gemm_tn()
    if "gemm" in which:
        for (M, tag) in [(16384, "S16K"), (131072, "S128K")]:
            bench_gemm(M, 7168, 5120, 1, tag + "/qkv")
            bench_gemm(M, 5120, 5120, 3, tag + "/o")
            bench_gemm(M, 13824, 5120, 5, tag + "/fc1_swiglu")
            bench_gemm(M, 5120, 13824, 3, tag + "/fc2")
        bench_gemm(8192, 8192, 8192, 0, "square8k")
        bench_gemm(4096, 4096, 4096, 0, "square4k")
        bench_gemm(64 * 1025, 3072, 1024, 1, "vit/qkv64f")
        bench_gemm(64 * 1025, 4096, 1024, 0, "vit/fc1")
        bench_gemm(64 * 1025, 1024, 4096, 0, "vit/fc2")
    if "attn" in which:
        for S in (4096, 16384, 32768, 131072):
            bench_attn(S, tag=f"llm{S}")
        bench_attn(1025, 16, 16, 64, False, 64, tag="vit64f")
    if "attn_bwd" in which:
        for S in (16384, 32768):
            bench_attn_bwd(S, tag=f"llm{S}")
    if "hbm" in which:
        bench_hbm()
