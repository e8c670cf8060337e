"""bench.py — prefill tokens/s/node of the Long-VITA hot path on MI355X.

    python bench.py --gpus N --steps K --warmup W
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...      (the same thing, ranks made by the caller)

Started WITHOUT a launcher (no WORLD_SIZE in the environment) and N > 1, this file launches its own N ranks under
torch.distributed.run on 127.0.0.1 (the way the reference's scripts start theirs:
R/scripts/megatron/qwen25/inference_qwen25_14b_intern_300m_server_cp.sh:96-181) and relays rank 0's ONE line.  A run that
cannot finish on the overlapped context-parallel schedule finishes on the plain one (one K / V message per layer, no side
streams, no own-chunks-first) and says so in the line's "degraded" field: inside a rank when the first step raises, and by a
second launch when the ranks of a self-launched run die or hang.

One "step" = one full prefill of the BASELINE.json metric's configuration: Long-VITA-128K —
a 506-frame synthetic video through InternViT-300M + pixel-shuffle projector, visual-token
scatter, the 48-layer 14B decoder at sequence 131072 and the logits-masked LM head producing the
next-token logits (one iteration of the reference's decode loop,
M/inference/text_generation/generation.py:123-205).  Context parallelism CP = N (zig-zag), TP = 1,
total work fixed as N grows ("strong" scaling).  Synthetic data, seeded random bf16 weights of
the real architecture; inputs are resident in HBM before the timed region.

Prints ONE JSON line on rank 0 (contract in the task statement) with these extra objects:
  roofline      dominant kernel = flash-attention forward: algorithmic FLOPs per launch
                (4*d*heads*visible (q,k) pairs on this rank) / mean launch time measured live with
                HIP events on the launch stream, against the 2.5 PFLOP/s dense bf16 MFMA peak;
  cpu_baseline  the host's time for the same workload from bounded samples (rank 0, every N): the CPU oracle ("port") and transformers'
                Qwen2 (the reference's HF path) are both timed; `value` is the faster one, both are stated;
  comm          (r04) what the exchange did, so that an N > 1 run can be read without a profiler: the communicator's backend and
                rank count, K / V all-gather messages and bytes per layer, the time the attention's stream actually WAITED for a
                gather (HIP events around every wait, mean per layer / total per prefill), the same all-gather timed alone after the
                timed region (ms, GB/s received per rank), the logits gather;
  cross_rank_check (N > 1) after the timed region: the logits of the 2 N positions the ranks marked (two per rank, sync_output's
                rows) against the SAME prefill run by rank 0 alone at CP = 1 — rel-L2, max-abs, argmax agreement.
"""
from __future__ import annotations

import argparse
import datetime
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)
# multi-process GPU work on this platform needs dmabuf IPC (RCCL's peer buffers): keep the variable set whatever launched us
os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")

import torch  # noqa: E402
import torch.distributed as dist  # noqa: E402

MFMA_BF16_PEAK_TFLOPS = 2500.0        # MI355X dense bf16 (MI355X_MICROARCH.md)


FIRST_CONTACT_TIMEOUT_S = 300          # N > 1: how long a rank waits in a collective / in the first-step vote for a peer that is gone


def committed_pmc_sets():
    """profiles/rNN_attn128k_pmc.json, newest round first — the PMC measurement `roofline.traffic` is read from (each is
    tools/pmc_to_json.py applied to the committed raw passes rNN_attn128k_pmc_raw.txt; tests/test_cpu_host.py re-derives the
    newest one)."""
    import glob
    import re
    found = [p for p in glob.glob(os.path.join(ROOT, "profiles", "r*_attn128k_pmc.json")) if re.fullmatch(r"r\d\d_attn128k_pmc\.json", os.path.basename(p))]
    return sorted(found, reverse=True)


def flops_per_token(seq, frames, cfg, vcfg):
    """algorithmic FLOPs of one prefill / seq."""
    lin = cfg.num_layers * 2 * (cfg.hidden * cfg.qkv_out + cfg.hidden * cfg.heads * cfg.head_dim
                                + cfg.hidden * 2 * cfg.ffn + cfg.ffn * cfg.hidden) * seq
    attn = cfg.num_layers * 4 * cfg.head_dim * cfg.heads * (seq * (seq + 1) // 2)
    s_v = vcfg.grid ** 2 + 1
    vit_frame = (vcfg.num_layers * (2 * s_v * (vcfg.hidden * 3 * vcfg.hidden + vcfg.hidden * vcfg.hidden
                                               + 2 * vcfg.hidden * vcfg.ffn) + 4 * s_v * s_v * vcfg.hidden)
                 + 2 * vcfg.grid ** 2 * 588 * vcfg.hidden
                 + 256 * 2 * (4 * vcfg.hidden * vcfg.hidden + vcfg.hidden * vcfg.llm_hidden))
    head = 2 * cfg.hidden * cfg.vocab
    return (lin + attn + frames * vit_frame + head) / seq


def cpu_baseline(seq: int, frames: int, fpt_workload: float):
    """The CPU oracle ("port") on the host cores, bounded samples of the same workload (tens of seconds):
      * 2 full-width decoder layers (hidden 5120, 40:8 heads, ffn 13824) at S = 2048: oracle.llm.decoder_layer;
      * single-layer causal attention (oracle.attention.core_attention, one kv group at a time) at S = 2K / 4K:
        BASELINE.md §3 B3 — its time follows b * S^2, and on a CPU the attention runs at a much lower FLOP rate than the
        GEMMs, so a FLOP-rate carry-over would flatter the host on this attention-dominated workload;
      * oracle.vit (24-layer InternViT + projector) on 2 frames;
      * BASELINE.md §3 B1, bounded: transformers' Qwen2ForCausalLM (what the reference's HF path wraps,
        H/models/long_vita_qwen2_intern/modeling_long_vita.py:227) at the 14B width, 2 layers, S = 2048.
    value = seq / (48 * (a * seq + b * seq^2) + frames * vit seconds per frame), a from the layer sample minus its own
    attention share, b from the attention samples — the host's time model evaluated at the benchmark sequence."""
    from oracle import glue, llm as ollm, vit as ovit
    from oracle.attention import core_attention
    S, L = 2048, 2
    cfg = ollm.LLMConfig(num_layers=L, vocab=64)
    p = ollm.init_llm_params(cfg, seed=1)
    h = (torch.randn(S, 1, cfg.hidden, generator=torch.Generator().manual_seed(0)) * 0.5).bfloat16()
    freqs = glue.rope_emb(S, glue.rope_inv_freq(cfg.head_dim, cfg.rope_theta))

    def attn_by_group(q, k, v):
        outs = [core_attention(q[:, :, g * cfg.qpg:(g + 1) * cfg.qpg], k[:, :, g:g + 1], v[:, :, g:g + 1], True).view(q.shape[0], 1, cfg.qpg, -1)
                for g in range(cfg.kv_groups)]
        return torch.cat(outs, 2).reshape(q.shape[0], 1, -1)

    t0 = time.perf_counter()
    with torch.no_grad():
        for lp in p["layers"]:
            h, _ = ollm.decoder_layer(h, lp, cfg, freqs, attn_by_group)
    t_layer = (time.perf_counter() - t0) / L
    qkv_out = (cfg.heads + 2 * cfg.kv_groups) * cfg.head_dim
    lin_flops_per_token = 2 * (cfg.hidden * qkv_out + cfg.hidden * cfg.heads * cfg.head_dim + 3 * cfg.hidden * cfg.ffn)
    flops_layer = S * lin_flops_per_token + 4 * cfg.head_dim * cfg.heads * (S * (S + 1) // 2)
    rate = flops_layer / t_layer
    attn = {}
    for s_a in (2048, 4096):
        g = torch.Generator().manual_seed(s_a)
        q = torch.randn(s_a, 1, cfg.heads, cfg.head_dim, generator=g).bfloat16()
        k = torch.randn(s_a, 1, cfg.kv_groups, cfg.head_dim, generator=g).bfloat16()
        v = torch.randn(s_a, 1, cfg.kv_groups, cfg.head_dim, generator=g).bfloat16()
        t0 = time.perf_counter()
        with torch.no_grad():
            attn_by_group(q, k, v)
        attn[s_a] = time.perf_counter() - t0
    b_coef = sum(attn[s_a] * s_a ** 2 for s_a in attn) / sum(float(s_a) ** 4 for s_a in attn)        # least squares through 0
    a_coef = max(t_layer - attn[S], 0.0) / S
    vcfg = ovit.ViTConfig()
    vp = ovit.init_vit_params(vcfg, seed=2)
    imgs = torch.randn(2, 3, 448, 448, generator=torch.Generator().manual_seed(3)).bfloat16()
    t0 = time.perf_counter()
    with torch.no_grad():
        ovit.vision_model(imgs, vp, vcfg)
    vit_s_per_frame = (time.perf_counter() - t0) / imgs.size(0)
    t_prefill = 48 * (a_coef * seq + b_coef * float(seq) ** 2) + frames * vit_s_per_frame
    # The reference's actual CPU path for the decoder: transformers' Qwen2ForCausalLM, 2 layers at the 14B width at
    # S = 2048 plus its attention call alone at 8K / 16K -> per-layer t(S) = a S + b S^2; the ViT share stays oracle.vit's (the HF
    # tower is the same torch ops).  `value` is the FASTER of the two host models (the eager port was 4 x slower on the linear part
    # and flattered the GPU); both are stated.
    hf, t_prefill_hf = {}, None
    try:
        import transformers
        hcfg = transformers.Qwen2Config(hidden_size=cfg.hidden, intermediate_size=cfg.ffn, num_hidden_layers=2,
                                        num_attention_heads=cfg.heads, num_key_value_heads=cfg.kv_groups, vocab_size=1024,
                                        max_position_embeddings=8192, rope_theta=cfg.rope_theta)
        hm = transformers.Qwen2ForCausalLM(hcfg).to(torch.bfloat16).eval()
        with torch.no_grad():
            hm(torch.zeros(1, 128, dtype=torch.long), use_cache=False, logits_to_keep=1)        # first-call set-up is not the host's pace
        ids = torch.randint(0, 1024, (1, S), generator=torch.Generator().manual_seed(5))
        t0 = time.perf_counter()
        with torch.no_grad():
            hm(ids, use_cache=False, logits_to_keep=1)
        t_layer_hf = (time.perf_counter() - t0) / 2
        # its attention is torch's scaled_dot_product_attention on [1, 40, S, 128] (kv heads repeated): b from S = 8K / 16K, where the
        # S^2 term dominates (at 2K-4K the call is overhead-bound and a two-point fit of whole layers gives b = 0)
        sd = {}
        for s_a in (8192, 16384):
            g = torch.Generator().manual_seed(s_a)
            q, k, v = (torch.randn(1, cfg.heads, s_a, cfg.head_dim, generator=g).bfloat16() for _ in range(3))
            t0 = time.perf_counter()
            with torch.no_grad():
                torch.nn.functional.scaled_dot_product_attention(q, k, v, is_causal=True)
            sd[s_a] = time.perf_counter() - t0
        b_hf = sum(sd[s_a] * s_a ** 2 for s_a in sd) / sum(float(s_a) ** 4 for s_a in sd)
        a_hf = max(t_layer_hf - b_hf * S * S, 0.0) / S
        t_prefill_hf = 48 * (a_hf * seq + b_hf * float(seq) ** 2) + frames * vit_s_per_frame
        hf = {"layers": 2, "seq": S, "seconds_per_layer": t_layer_hf, "sdpa_seconds_by_seq": sd, "linear_s_per_token_per_layer": a_hf,
              "attention_s_per_token2": b_hf, "tokens_per_s_at_benchmark_seq": seq / t_prefill_hf, "prefill_seconds_modelled": t_prefill_hf,
              "tokens_per_s_extrapolated_to_48_layers_at_2048": S / (t_layer_hf * 48), "transformers": transformers.__version__}
        del hm
    except Exception as e:  # noqa: BLE001
        hf = {"error": f"{type(e).__name__}: {e}"}
    port_value = seq / t_prefill
    use_hf = t_prefill_hf is not None and t_prefill_hf < t_prefill
    sample_port = (f"{L} of 48 full-width decoder layers at S={S} (oracle.llm.decoder_layer): {t_layer:.2f} s per layer = "
                   f"{rate / 1e12:.2f} TFLOP/s; single-layer causal attention (oracle.attention.core_attention) "
                   + ", ".join(f"{attn[s_a]:.2f} s @ {s_a}" for s_a in attn)
                   + f" -> b = {b_coef:.3e} s/token^2, a = {a_coef:.3e} s/token per layer; oracle.vit on 2 frames: "
                   f"{vit_s_per_frame:.2f} s/frame; {seq} / (48 (a S + b S^2) + {frames} frames) = {seq} tokens / {t_prefill:.0f} s")
    if use_hf:
        sample = ("transformers Qwen2ForCausalLM (what the reference's HF path wraps, H/models/long_vita_qwen2_intern/modeling_long_vita.py:227), "
                  f"2 of 48 layers at the 14B width: {hf['seconds_per_layer']:.2f} s per layer at S = {S}; its attention (torch sdpa, 40 heads, causal) "
                  + ", ".join(f"{hf['sdpa_seconds_by_seq'][s_a]:.2f} s @ {s_a}" for s_a in hf['sdpa_seconds_by_seq'])
                  + f" -> a = {hf['linear_s_per_token_per_layer']:.3e} s/token, b = {hf['attention_s_per_token2']:.3e} s/token^2 per layer; "
                  f"oracle.vit on 2 frames: {vit_s_per_frame:.2f} s/frame; value = {seq} tokens / {t_prefill_hf:.0f} s (model evaluated at the "
                  f"benchmark sequence, not run).  The eager oracle port on the same host: {port_value:.3f} tokens/s [{sample_port}]")
    else:
        sample = sample_port + " (model evaluated at the benchmark sequence, not run)"
    return {"value": seq / t_prefill_hf if use_hf else port_value, "unit": "tokens/s", "cores": torch.get_num_threads(),
            "kind": "reference" if use_hf else "port", "sample": sample,
            "port_value": port_value, "reference_transformers_value": None if t_prefill_hf is None else seq / t_prefill_hf,
            "decoder_layer_s_at_2048": t_layer, "decoder_layer_tflops_at_2048": rate / 1e12,
            "attention_seconds_by_seq": attn, "attention_s_per_token2": b_coef, "linear_s_per_token_per_layer": a_coef,
            "vit_s_per_frame": vit_s_per_frame, "flop_rate_carry_over_tokens_per_s": rate / fpt_workload,
            "hf_transformers_qwen2": hf}


def comm_report(world: int, steps: int, cfg, seq: int, comm_log, dev, selftest: bool = False):
    """The `comm` object of the line (module docstring).  Collective on every rank (the isolated all-gather is one).
    selftest: `--dry-run` under torch.distributed.run with ONE rank walks the N > 1 code (a world of one) so that it has run somewhere
    before the driver's multi-GPU node runs it."""
    from long_vita_amd import ops, parallel_state as mpu
    if world == 1 and not selftest:
        return {"backend": None, "ranks": 1, "kv_messages_per_layer": 0, "kv_bytes_sent_per_layer_per_rank": 0,
                "exposed_wait_ms_per_layer": 0.0, "exposed_wait_ms_per_prefill": 0.0,
                "note": "one rank: no exchange on the path (the forced-CP parity pass runs outside the timed region)"}
    group = mpu.get_context_parallel_group()
    s_l = seq // world
    n_msg = ops.cp_kv_split(cfg.kv_groups, cfg.heads, s_l)
    waits = [e["wait"][0].elapsed_time(e["wait"][1]) for e in comm_log]
    sent = sum(e["bytes"] for e in comm_log)
    layers = max(cfg.num_layers * steps, 1)
    msg_bytes = 2 * s_l * (cfg.kv_groups // n_msg) * cfg.head_dim * 2
    # the same message timed alone: blocking all-gathers bracketed by HIP events on the stream they are ordered on
    send = torch.empty(msg_bytes // 2, dtype=torch.bfloat16, device=dev).normal_()
    recv = torch.empty(world * send.numel(), dtype=torch.bfloat16, device=dev)
    for _ in range(2):
        dist.all_gather_into_tensor(recv, send, group=group)
    torch.cuda.synchronize()
    reps = 10
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(reps):
        dist.all_gather_into_tensor(recv, send, group=group)
    b.record()
    torch.cuda.synchronize()
    iso_ms = a.elapsed_time(b) / reps
    t = torch.tensor([iso_ms, sum(waits)], dtype=torch.float64, device=dev)
    dist.all_reduce(t, op=dist.ReduceOp.MAX, group=group)
    iso_max, wait_max = float(t[0]), float(t[1])
    return {"backend": dist.get_backend(group), "ranks": dist.get_world_size(group), "kv_messages_per_layer": n_msg,
            "kv_message_bytes": msg_bytes, "kv_bytes_sent_per_layer_per_rank": sent / layers,
            "kv_bytes_received_per_layer_per_rank": (world - 1) * sent / layers,
            "gathers_logged_rank0": len(comm_log),
            "exposed_wait_ms_per_layer": sum(waits) / layers, "exposed_wait_ms_per_prefill": sum(waits) / max(steps, 1),
            "exposed_wait_ms_per_prefill_max_over_ranks": wait_max / max(steps, 1),
            "isolated_all_gather_ms": iso_ms, "isolated_all_gather_ms_max_over_ranks": iso_max,
            "isolated_all_gather_gbps_received_per_rank": (world - 1) * msg_bytes / (iso_ms * 1e-3) / 1e9,
            "how": "exposed wait = HIP events around each point where an attention stream waits for its K / V gather (rank 0); isolated = "
                   f"{reps} blocking all_gather_into_tensor calls of one message, HIP events on the current stream, after the timed region"}


def cross_rank_check(model, tokens, seq, ext, world, rank):
    """N > 1, outside the timed region: the CP = N prefill's logits at the 2 N marked positions (row j of sync_output = global position
    j * S / (2 N) + (S - 1) % (S / (2 N)) at context_length = S) against the same prefill run by rank 0 ALONE at CP = 1 on the same
    inputs (the other ranks wait at a barrier)."""
    from long_vita_amd import generation, parallel_state as mpu
    position_ids = torch.arange(seq, dtype=torch.long, device=tokens.device).unsqueeze(0)
    t2, p2, e2 = generation.get_batch_on_this_cp_rank(tokens, position_ids, ext)
    mask, _ = generation.build_logit_mask(t2, seq, False)
    rows = generation.sync_output(model(t2, p2, None, external_inputs=e2, logit_mask=mask))          # [1, 2 N, V]
    half = seq // (2 * world)
    pos = [j * half + (seq - 1) % half for j in range(2 * world)]
    res = None
    if rank == 0:
        state = (mpu.get_context_parallel_world_size(), mpu.get_context_parallel_rank(), mpu.get_context_parallel_group())
        try:
            mpu.set_context_parallel_state(1, 0, None)
            m1 = torch.zeros(1, seq, dtype=torch.bool, device=tokens.device)
            m1[0, pos] = True
            alone = model(tokens, position_ids, None, external_inputs=ext, logit_mask=m1)             # [1, 2 N, V]
            a, b = rows.float(), alone.float()
            res = {"what": f"logits at the {2 * world} marked positions: CP = {world} vs the same prefill on rank 0 alone (CP = 1)",
                   "positions": pos, "rel_l2": float((a - b).norm() / b.norm()), "max_abs": float((a - b).abs().max()),
                   "argmax_equal": int((a.argmax(-1) == b.argmax(-1)).sum()), "rows": 2 * world}
        except Exception as e:  # noqa: BLE001 — the other ranks are waiting at the barrier below
            res = {"error": f"{type(e).__name__}: {e}"}
        finally:
            mpu.set_context_parallel_state(*state)
    model._ws = {}
    dist.barrier()
    return res


DEGRADED_ENV = {"VITA_CP_STREAMS": "0", "VITA_CP_LOCAL_FIRST": "0", "VITA_CP_KV_SPLIT": "1"}


def launch_command(gpus: int, port: int, argv):
    """The driver's own command form for N ranks on one node (task statement), with this file and its arguments."""
    return [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(gpus), "--master-addr", "127.0.0.1",
            "--master-port", str(port), os.path.abspath(__file__)] + list(argv)


def _free_port() -> int:
    import socket
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _run_ranks(cmd, env, timeout):
    """One launch: (return code or None when killed on the time limit, captured stdout).  stderr goes straight through.  The ranks get
    their own process group so that a hung launch can be ended as a whole (by its pgid, nothing else)."""
    import signal
    import subprocess
    p = subprocess.Popen(cmd, env=env, stdout=subprocess.PIPE, text=True, start_new_session=True)
    try:
        out, _ = p.communicate(timeout=timeout)
        return p.returncode, out
    except subprocess.TimeoutExpired:
        try:
            os.killpg(p.pid, signal.SIGKILL)
        except ProcessLookupError:
            pass
        out, _ = p.communicate()
        return None, out


def self_launch(gpus: int, argv, steps: int, warmup: int, run=_run_ranks, environ=None) -> int:
    """`python bench.py --gpus N` with no launcher around it: start the N ranks, relay rank 0's JSON line, return the exit code.
    If the launch dies or exceeds its time limit before a line appears, launch ONCE more on the plain exchange schedule
    (DEGRADED_ENV) with the reason handed down in VITA_BENCH_DEGRADED, so that a slower number comes out instead of none."""
    environ = dict(os.environ if environ is None else environ)
    environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    environ.pop("VITA_BENCH_FORCE_SPAWN", None)                       # the ranks must not launch again
    environ["VITA_BENCH_SELF_LAUNCHED"] = "1"
    # generous: model build + (warm-up + timed + parity / diagnostics) prefills at the one-GPU pace + the CPU baseline
    limit = float(environ.get("VITA_BENCH_LAUNCH_TIMEOUT", 900 + 25 * (steps + warmup + 4)))

    def attempt(env):
        rc, out = run(launch_command(gpus, _free_port(), argv), env, limit)
        lines = [ln for ln in (out or "").splitlines() if ln.startswith("{") and '"metric"' in ln]
        return rc, lines, out

    rc, lines, out = attempt(environ)
    if rc == 0 and lines:
        print(lines[-1], flush=True)
        return 0
    reason = (f"first launch of {gpus} ranks " + ("exceeded its time limit" if rc is None else f"exited with code {rc}")
              + ("" if lines else " before printing a line") + "; re-run on the plain exchange schedule")
    print(f"bench.py: {reason}", file=sys.stderr, flush=True)
    sys.stderr.write((out or "")[-4000:])
    rc2, lines2, out2 = attempt(dict(environ, VITA_BENCH_DEGRADED=reason, **DEGRADED_ENV))
    if lines2:
        print(lines2[-1], flush=True)
        return 0 if rc2 == 0 else (rc2 or 1)
    sys.stderr.write((out2 or "")[-4000:])
    return rc2 or 1


def degrade_exchange(model) -> None:
    """The plain context-parallel schedule on a live model: one K / V all-gather per layer, attention on the current stream after
    it, no own-chunks-first split (what DEGRADED_ENV selects at construction)."""
    os.environ.update(DEGRADED_ENV)
    att = model.core_attention
    att.local_first, att.split_streams = False, False
    att._kv_gather, att._o_remote = None, None
    model._ws = {}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=2)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--seq", type=int, default=131072, help="debug only; the metric is quoted at 131072")
    ap.add_argument("--layers", type=int, default=48, help="debug only; the metric needs all 48 layers")
    ap.add_argument("--vit-layers", type=int, default=24, help="debug only")
    ap.add_argument("--frames", type=int, default=-1, help="debug only; default fills the sequence (506 @128K)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-parity-check", action="store_true")
    ap.add_argument("--dry-run", action="store_true", help="plumbing check: tiny model (2 + 2 layers, seq 4096, 8 frames); NOT a measurement")
    args = ap.parse_args()

    if args.dry_run:
        args.seq, args.layers, args.vit_layers, args.frames, args.no_cpu_baseline = 4096, 2, 2, 8, True
    # test hooks (VITA_BENCH_FORCE_SPAWN, VITA_BENCH_INJECT_FAILURE) are honoured under --dry-run only: the measured entry point has none
    force_spawn = args.dry_run and os.environ.get("VITA_BENCH_FORCE_SPAWN", "0") == "1"       # tests: walk the self-launch with N = 1
    if "WORLD_SIZE" not in os.environ and (args.gpus > 1 or force_spawn):
        raise SystemExit(self_launch(args.gpus, sys.argv[1:], args.steps, args.warmup))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world != args.gpus:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}: the launcher's --nproc-per-node and --gpus must agree")
    # VITA_BENCH_BACKEND=gloo-staged (diagnostic, --dry-run only): N processes on whatever devices exist, device tensors exchanged through
    # host memory by gloo (tools/gloo_staging.py) — walks the N > 1 code on a one-GPU box, where RCCL refuses two ranks on one device
    staged = os.environ.get("VITA_BENCH_BACKEND", "nccl") == "gloo-staged"
    if staged and not args.dry_run:
        raise SystemExit("VITA_BENCH_BACKEND=gloo-staged is a plumbing check: it is accepted together with --dry-run only")
    if staged:
        local_rank %= max(torch.cuda.device_count(), 1)
    torch.cuda.set_device(local_rank)
    dev = f"cuda:{local_rank}"
    selftest = "MASTER_ADDR" in os.environ and (args.dry_run or os.environ.get("VITA_BENCH_SELF_LAUNCHED") == "1")
    ctl = None
    if world > 1 or selftest:
        if staged:
            dist.init_process_group("gloo")
            sys.path.insert(0, os.path.join(ROOT, "tools"))
            import gloo_staging
            gloo_staging.install()
        else:
            dist.init_process_group("nccl", device_id=torch.device(dev), timeout=datetime.timedelta(seconds=FIRST_CONTACT_TIMEOUT_S))
        # a control plane that does not ride on the thing being tested: the "did every rank get through the first step" vote.
        # The vote covers failures EVERY rank sees (a collective that raises everywhere).  If ONE rank raises while the others sit in that
        # step's all-gather, they never reach the vote: RCCL's watchdog ends them after FIRST_CONTACT_TIMEOUT_S, the rank waiting in the vote
        # gives up after the same time (the gloo group's timeout), and the self-launcher's ONE relaunch on the plain schedule takes over
        # (a run started by an outside launcher has no second attempt: its ranks exit non-zero) — ADVICE r05.
        ctl = dist.new_group(backend="gloo", timeout=datetime.timedelta(seconds=FIRST_CONTACT_TIMEOUT_S))

    from long_vita_amd import generation, gpt_vl_model, lib, parallel_state as mpu, synthetic, vision
    lib.load(allow_build=False)                      # the HIP path or nothing
    mpu.initialize_model_parallel()

    cfg = gpt_vl_model.GPTConfig(num_layers=args.layers)
    vcfg = vision.VisionConfig(num_layers=args.vit_layers)
    seq = args.seq
    frames = synthetic.frames_for_seq(seq, tail_text=512) if args.frames < 0 else args.frames
    vit = vision.MegatronVisionModel.random_init(vcfg, seed=4321, device=dev)
    model = gpt_vl_model.GPTVLModel.random_init(cfg, seed=1234, device=dev, external_feature_model=vit)
    tokens, ext = synthetic.make_request(seq, frames, seed=1234, device=dev)
    torch.cuda.synchronize()

    def step():
        return generation.prefill_step(model, tokens, seq, ext, reference_compat=False)

    def fence():
        if dist.is_initialized():
            dist.barrier()
        torch.cuda.synchronize()

    # first contact with the exchange (N > 1): if ANY rank's first step raises, every rank drops to the plain schedule together
    degraded = os.environ.get("VITA_BENCH_DEGRADED")
    warm_done = 0
    if ctl is not None:
        ok, why = 1, ""
        try:
            inject = os.environ.get("VITA_BENCH_INJECT_FAILURE") if (degraded is None and args.dry_run) else None      # tests only
            if inject == "exit":
                os._exit(3)                         # a rank that dies: the self-launcher's second attempt
            if inject == "1":
                raise RuntimeError("injected first-step failure")
            step()
            torch.cuda.synchronize()
        except Exception as e:  # noqa: BLE001
            ok, why = 0, f"{type(e).__name__}: {e}"[:300]
        vote = torch.tensor([ok], dtype=torch.int32)
        dist.all_reduce(vote, op=dist.ReduceOp.MIN, group=ctl)
        if int(vote.item()) == 0:
            reasons = [None] * world
            dist.all_gather_object(reasons, why, group=ctl)
            first = next((f"rank {i}: {w}" for i, w in enumerate(reasons) if w), "unknown")
            degraded = f"first step on the overlapped exchange failed ({first}); plain schedule from there on"
            degrade_exchange(model)
            step()                                  # the plain schedule must get through, or the run ends here with the error
            torch.cuda.synchronize()
        warm_done = 1
    for _ in range(max(args.warmup - warm_done, 0)):
        step()
    model.attn_events = []
    if world > 1:
        model.core_attention.comm_log = []
    fence()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        out = step()
    fence()
    dt = time.perf_counter() - t0
    if world > 1:
        t = torch.tensor([dt], dtype=torch.float64, device=dev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = float(t.item())
    assert torch.isfinite(out.float()).all()
    comm_log = model.core_attention.comm_log or []
    model.core_attention.comm_log = None
    # diagnostics of the exchange: never allowed to cost the measured line (every rank takes the same path through the collectives)
    try:
        comm = comm_report(world, args.steps, cfg, seq, comm_log, dev, selftest=selftest and dist.is_initialized())
    except Exception as e:  # noqa: BLE001
        comm = {"error": f"{type(e).__name__}: {e}"}
    cross = None
    if world > 1 and not args.no_parity_check:
        try:
            cross = cross_rank_check(model, tokens, seq, ext, world, rank)
        except Exception as e:  # noqa: BLE001
            cross = {"error": f"{type(e).__name__}: {e}"}

    # once, outside the timed region (N = 1): the same prefill through the context-parallel code path (K/V pack, all-gather
    # messages, zig-zag chunk tables, logits gather) must give the plain path's logits
    parity = None
    attn_events = model.attn_events
    model.attn_events = None
    if world == 1 and not args.no_parity_check:
        model.force_cp_path = True
        out_cp = step()
        model.force_cp_path = False
        torch.cuda.synchronize()
        a, b_ = out_cp.float(), out.float()
        parity = {"what": "logits of the forced context-parallel path vs the plain path, same inputs, outside the timed region",
                  "rel_l2": float((a - b_).norm() / b_.norm()), "max_abs": float((a - b_).abs().max()),
                  "argmax_equal": bool((a.argmax(-1) == b_.argmax(-1)).all())}
        assert parity["rel_l2"] < 2e-2, parity

    # dominant kernel: flash attention forward, one launch per layer per step on this rank
    ev_ms = [a.elapsed_time(b) for a, b in attn_events]
    attn_ms = sum(ev_ms) / max(len(ev_ms), 1)
    pairs = seq * (seq + 1) // 2 / world
    attn_flops = 4 * cfg.head_dim * cfg.heads * pairs
    achieved = attn_flops / (attn_ms * 1e-3) / 1e12 if attn_ms > 0 else 0.0

    # HBM traffic of that kernel: PMC counters cannot be read from inside this process; when a
    # rocprofv3 --pmc measurement of the same launch shape is committed under profiles/, report it
    traffic, traffic_src = None, None
    for path in committed_pmc_sets():                      # the newest committed measurement of this launch shape
        try:
            pmc = json.load(open(path))
            if pmc["seq"] == seq and pmc["n_gpus"] == world:
                traffic = pmc["hbm_bytes_per_launch"]
                traffic_src = f"profiles/{os.path.basename(path)} (rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE, offline)"
                break
        except (OSError, KeyError, ValueError):
            continue

    ms_per_step = dt / args.steps * 1e3
    value = seq / (dt / args.steps)
    fpt = flops_per_token(seq, frames, cfg, vcfg)
    name = {16384: "Long-VITA-16K", 131072: "Long-VITA-128K", 1048576: "Long-VITA-1M"}.get(seq, f"Long-VITA (seq {seq})")
    line = {
        "metric": "prefill tokens/sec/node (ViT+LLM) at seq=128K" if seq == 131072 else f"prefill tokens/sec/node (ViT+LLM) at seq={seq}",
        "value": value, "unit": "tokens/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
        "ms_per_step": ms_per_step, "higher_is_better": True, "scaling": "strong", "vs_baseline": None,
        "dtype": "bf16", "data": "synthetic",
        "config": {"workload": f"{name} prefill: {frames}-frame synthetic video (InternViT-300M + projector) + "
                               f"Qwen2.5-14B decoder ({cfg.num_layers} layers), seq {seq}, logits-masked LM head",
                   "seq_len": seq, "frames": frames, "global_batch": 1, "parallelism": f"cp{world}", "dry_run": bool(args.dry_run),
                   "weights": "seeded random bf16 (N(0,0.02))",
                   "algorithmic_gflop_per_token": fpt / 1e9,
                   "end_to_end_tflops_per_gpu": fpt * value / world / 1e12,
                   "end_to_end_frac_of_mfma_peak": fpt * value / world / 1e12 / MFMA_BF16_PEAK_TFLOPS,
                   # measured on rank 0, to set against the byte model of DESIGN.md 6.1 (weights + workspace + frames)
                   "peak_hbm_gb_rank0": torch.cuda.max_memory_allocated() / 2 ** 30},
        "roofline": {"bound": "mfma", "kernel": "flash_fwd64_kernel (d = 128, causal; 4 waves x 64 rows)" if os.environ.get("VITA_ATTN64", "1") != "0" else "flash_fwd_kernel<128, causal>", "achieved": achieved,
                     "peak": MFMA_BF16_PEAK_TFLOPS, "unit": "TFLOP/s", "frac": achieved / MFMA_BF16_PEAK_TFLOPS,
                     "traffic": traffic, "traffic_source": traffic_src, "launches_timed": len(ev_ms), "ms_per_launch": attn_ms,
                     "flop_per_launch": attn_flops},
    }
    if degraded:
        line["degraded"] = degraded
    if staged:
        line["transport"] = "gloo-staged (diagnostic: device tensors through host memory; not a measurement)"
    if parity is not None:
        line["parity_check"] = parity
    line["comm"] = comm
    if cross is not None:
        line["cross_rank_check"] = cross
    if rank == 0:
        if not args.no_cpu_baseline:
            try:
                line["cpu_baseline"] = cpu_baseline(seq, frames, fpt)
            except Exception as e:  # noqa: BLE001 — a reported baseline must not cost the measured line
                line["cpu_baseline"] = {"value": None, "unit": "tokens/s", "cores": torch.get_num_threads(), "kind": "port",
                                        "sample": f"failed: {type(e).__name__}: {e}"}
        else:
            line["cpu_baseline"] = None
        print(json.dumps(line), flush=True)
    if dist.is_initialized():
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
