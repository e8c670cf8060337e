"""Comm / compute co-residency probe (VERDICT r2 "next round" 4), one MI355X.

Question: while `flash_fwd64_kernel` — one 512-register workgroup per CU, 640 workgroups per kv-head split at the CP = 8
per-rank geometry of the 128K prefill (S_l = 16384 queries x 131072 gathered keys, 10 : 2 heads) — is running or queued, does a
communication kernel shaped like RCCL's (few 256-thread channels, few registers) get CUs, when, and what does it cost the
attention?  And a copy engine (SDMA) instead?

Stream A: 4 attention launches (the 4 kv-head splits of one layer, ~5 ms each).  Stream B, per scenario, 4 messages of 16.8 MB (one
rank's K/V shard of one split):
  rccl280-...   the same channel kernels with RCCL's REAL footprint (rcclGenericKernel of this image's librccl.so: 256 threads,
                280 registers, 19.7 KB LDS = one wave per SIMD, so a channel needs a CU with no attention workgroup on it);
  cu<N>/first   N-channel light (12-register) CU copy kernels (tools/hwprobe/comm_overlap.hip) enqueued BEFORE the attention (forward_cp's order: all
                gathers are issued up front);
  cu<N>/late    the same enqueued 2 ms AFTER the first attention launch (the chip is full of attention workgroups);
  ...paced      each channel stays resident 0.4 ms per message (a transfer paced by an xGMI link, not by HBM);
  d2d           hipMemcpyAsync device -> device (the runtime's engine choice);
  h2d / d2h     hipMemcpyAsync against pinned host memory (SDMA engines over PCIe: the engines a peer copy over xGMI would use).
Every kernel stamps the device-wide 100 MHz clock, so "when did message j start / end relative to the attention" is measured
on the device, not inferred from the host.  Prints one line per scenario; writes gpurun_out/r03_hwprobe_comm_overlap.txt."""
import ctypes as C
import json
import os
import statistics
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402

from long_vita_amd import lib, ops  # noqa: E402

DEV = "cuda:0"
OUT = os.path.join(ROOT, "gpurun_out")
os.makedirs(OUT, exist_ok=True)
LOG = open(os.path.join(OUT, "r03_hwprobe_comm_overlap.txt"), "w")


def say(s):
    print(s, flush=True)
    LOG.write(s + "\n")
    LOG.flush()


lib.load(allow_build=False)
P = C.CDLL(os.path.join(ROOT, "tools", "hwprobe", "bin", "libcomm_overlap.so"))
P.probe_channel_copy.argtypes = [C.c_void_p, C.c_void_p, C.c_int64, C.c_int, C.c_void_p, C.c_uint64, C.c_int, C.c_void_p]
P.probe_stamp.argtypes = [C.c_void_p, C.c_void_p]
P.probe_memcpy_async.argtypes = [C.c_void_p, C.c_void_p, C.c_int64, C.c_int, C.c_void_p]

# ---- the attention of one rank of CP = 8 at 128K, one kv-head split per launch ---------------------------------------------------
cp, r, S, hg, G, D = 8, 3, 131072, 2, 5, 128
c = S // (2 * cp)
s_l = 2 * c
gen = torch.Generator(device=DEV).manual_seed(1)
q = torch.randn(1, s_l, hg * G, D, generator=gen, device=DEV).bfloat16()
rows = torch.randn(cp * 2 * s_l, hg, D, generator=gen, device=DEV).bfloat16()
o = torch.empty_like(q)
kv_gid, kv_row = [], []
for p in range(cp):
    kv_gid += [p, 2 * cp - 1 - p]
    kv_row += [p * 2 * s_l, p * 2 * s_l + c]
own = [r, 2 * cp - 1 - r]
N_ATT = 4


def attention():
    for _ in range(N_ATT):
        ops.flash_attn(q, rows.unsqueeze(0), rows[s_l:].unsqueeze(0), causal=True, chunk_len=c, q_chunk_gid=own, kv_chunk_gid=kv_gid,
                       kv_chunk_row=kv_row, out=o)


MSG = 2 * s_l * hg * D * 2            # bytes of one rank's packed K/V shard of one split: 16.8 MB
N_MSG = 4
src = torch.empty(N_MSG, MSG, dtype=torch.uint8, device=DEV).random_(0, 255)
dst = torch.empty_like(src)
host = torch.empty(N_MSG, MSG, dtype=torch.uint8).pin_memory()
A, B = torch.cuda.Stream(), torch.cuda.Stream()
stamps_a = torch.zeros(2, dtype=torch.int64, device=DEV)
stamps_b = torch.zeros(N_MSG, 2 * 64, dtype=torch.int64, device=DEV)


def stamp(slot, stream):
    P.probe_stamp(stamps_a[slot:].data_ptr(), stream.cuda_stream)


# ---- clock rate of s_memrealtime against HIP events ------------------------------------------------------------------------------
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
with torch.cuda.stream(A):
    e0.record(); stamp(0, A)
    attention()
    stamp(1, A); e1.record()
torch.cuda.synchronize()
TICKS_PER_MS = float(stamps_a[1] - stamps_a[0]) / e0.elapsed_time(e1)
say(f"clock: {TICKS_PER_MS / 1e3:.2f} MHz; message = {MSG / 1e6:.1f} MB x {N_MSG}; attention = {N_ATT} launches of {hg * G} heads x {s_l // 256} row tiles = "
    f"{hg * G * s_l // 256} workgroups each (S_l = {s_l}, {2 * cp} key chunks of {c})")


def run(kind, n_ch=0, late=False, pace_ms=0.0, heavy=0):
    """One scenario -> dict(attn_ms, msgs=[(start_ms, end_ms) relative to the attention's first instruction])."""
    stamps_b.zero_()
    ea, eb0, eb1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    ea1 = torch.cuda.Event(enable_timing=True)
    msg_ev = []
    torch.cuda.synchronize()

    def messages():
        with torch.cuda.stream(B):
            eb0.record()
            for j in range(N_MSG):
                if kind == "cu":
                    P.probe_channel_copy(dst[j].data_ptr(), src[j].data_ptr(), MSG, n_ch, stamps_b[j].data_ptr(),
                                         int(pace_ms * TICKS_PER_MS), heavy, B.cuda_stream)
                elif kind == "d2d":
                    P.probe_memcpy_async(dst[j].data_ptr(), src[j].data_ptr(), MSG, 0, B.cuda_stream)
                elif kind == "h2d":
                    P.probe_memcpy_async(dst[j].data_ptr(), host[j].data_ptr(), MSG, 1, B.cuda_stream)
                elif kind == "d2h":
                    P.probe_memcpy_async(host[j].data_ptr(), src[j].data_ptr(), MSG, 2, B.cuda_stream)
                ev = torch.cuda.Event(enable_timing=True)
                ev.record()
                msg_ev.append(ev)
            eb1.record()

    def attn():
        with torch.cuda.stream(A):
            ea.record(); stamp(0, A)
            attention()
            stamp(1, A); ea1.record()

    if kind == "none":
        attn()
    elif late:
        attn()
        time.sleep(0.002)
        messages()
    else:
        messages()
        attn()
    torch.cuda.synchronize()
    res = {"attn_ms": float(stamps_a[1] - stamps_a[0]) / TICKS_PER_MS, "attn_ms_events": ea.elapsed_time(ea1)}
    if kind == "cu":
        t0 = int(stamps_a[0])
        st = stamps_b[:, : 2 * n_ch].view(N_MSG, n_ch, 2)
        res["msgs"] = [((int(st[j, :, 0].min()) - t0) / TICKS_PER_MS, (int(st[j, :, 1].max()) - t0) / TICKS_PER_MS,
                        (int(st[j, :, 0].max()) - t0) / TICKS_PER_MS) for j in range(N_MSG)]
    elif kind != "none":
        prev = eb0
        res["msgs"] = []
        for ev in msg_ev:
            res["msgs"].append((ea.elapsed_time(prev), ea.elapsed_time(ev)))
            prev = ev
    return res


def report(name, **kw):
    runs = [run(**kw) for _ in range(3)]
    a = statistics.median(x["attn_ms"] for x in runs)
    line = f"{name:34s} attention {a:7.3f} ms ({a / BASE * 100 - 100:+5.1f} %)"
    mid = sorted(runs, key=lambda x: x["attn_ms"])[1]
    if "msgs" in mid:
        if kw.get("kind") == "cu":
            line += "  messages [first channel starts, last channel ends | last channel starts] ms after attention start: " + "  ".join(
                f"[{m[0]:+.2f}, {m[1]:+.2f} | {m[2]:+.2f}]" for m in mid["msgs"])
        else:
            line += "  messages [enqueue-ordered start, end] ms after attention start: " + "  ".join(f"[{m[0]:+.2f}, {m[1]:+.2f}]" for m in mid["msgs"])
    say(line)
    RESULTS[name] = {"attn_ms": a, "delta_pct": a / BASE * 100 - 100, "msgs": mid.get("msgs")}


RESULTS = {}
BASE = 1.0
run("none")
BASE = statistics.median(run("none")["attn_ms"] for _ in range(5))
say(f"{'attention alone':34s} attention {BASE:7.3f} ms")
RESULTS["attention alone"] = {"attn_ms": BASE}
# copies alone (nothing on stream A): how long does one message take by itself?
for n_ch in (8, 32):
    stamps_b.zero_()
    torch.cuda.synchronize()
    with torch.cuda.stream(B):
        for j in range(N_MSG):
            P.probe_channel_copy(dst[j].data_ptr(), src[j].data_ptr(), MSG, n_ch, stamps_b[j].data_ptr(), 0, 0, B.cuda_stream)
    torch.cuda.synchronize()
    st = stamps_b[:, : 2 * n_ch].view(N_MSG, n_ch, 2)
    d = [(int(st[j, :, 1].max()) - int(st[j, :, 0].min())) / TICKS_PER_MS for j in range(N_MSG)]
    say(f"cu{n_ch} copy alone: {statistics.median(d):.3f} ms per {MSG / 1e6:.1f} MB message = {MSG / statistics.median(d) / 1e6:.0f} GB/s read + the same written")
for n_ch in (8, 16, 32):
    report(f"cu{n_ch}/first", kind="cu", n_ch=n_ch)
    report(f"cu{n_ch}/late", kind="cu", n_ch=n_ch, late=True)
report("cu16/first paced 0.4 ms", kind="cu", n_ch=16, pace_ms=0.4)
report("cu16/late paced 0.4 ms", kind="cu", n_ch=16, late=True, pace_ms=0.4)
report("cu16/late paced 2 ms", kind="cu", n_ch=16, late=True, pace_ms=2.0)
# the same with RCCL's real footprint (280 registers, 19.7 KB LDS per 256-thread channel: one wave per SIMD, needs a whole CU)
for n_ch in (16, 32):
    report(f"rccl280-ch{n_ch}/first", kind="cu", n_ch=n_ch, heavy=1)
    report(f"rccl280-ch{n_ch}/late", kind="cu", n_ch=n_ch, late=True, heavy=1)
report("rccl280-ch16/first paced 0.4 ms", kind="cu", n_ch=16, pace_ms=0.4, heavy=1)
report("rccl280-ch16/late paced 0.4 ms", kind="cu", n_ch=16, late=True, pace_ms=0.4, heavy=1)
for kind in ("d2d", "h2d", "d2h"):
    report(f"{kind}/first", kind=kind)
    report(f"{kind}/late", kind=kind, late=True)
json.dump(RESULTS, open(os.path.join(OUT, "r03_hwprobe_comm_overlap.json"), "w"), indent=1)
