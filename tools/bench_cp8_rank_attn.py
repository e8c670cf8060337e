"""One rank's attention forward at the CP = 8 geometry of the 128K prefill (S_l = 16384 local queries, zig-zag chunks r and 15 - r,
against the 131072 gathered keys; 40 : 8 heads) as the kv-head splits launch it, on ONE GPU without communication:
  n_split launches of 8 / n_split kv heads (2560 / n_split workgroups each for 256 CUs) on one stream, on n_split streams, and the
  whole layer as ONE launch (the lower bound: 10 full rounds).
Answers: what does splitting the layer into per-gather launches cost, and do separate streams give it back?
Prints JSON lines; appends to gpurun_out/r03_cp8_rank_attn.jsonl."""
import json
import os
import statistics
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402

from long_vita_amd import lib, ops  # noqa: E402

DEV = "cuda:0"
lib.load(allow_build=False)
os.makedirs(os.path.join(ROOT, "gpurun_out"), exist_ok=True)
LOG = open(os.path.join(ROOT, "gpurun_out", "r03_cp8_rank_attn.jsonl"), "a")
cp, S, Hq, Hkv, D = 8, 131072, 40, 8, 128
c = S // (2 * cp)
s_l = 2 * c
G = Hq // Hkv
gen = torch.Generator(device=DEV).manual_seed(1)
q = torch.randn(1, s_l, Hq, D, generator=gen, device=DEV).bfloat16()
o = torch.empty_like(q)
kv_gid, kv_row = [], []
for p in range(cp):
    kv_gid += [p, 2 * cp - 1 - p]
    kv_row += [p * 2 * s_l, p * 2 * s_l + c]


def emit(**kw):
    s = json.dumps(kw)
    print(s, flush=True)
    LOG.write(s + "\n"); LOG.flush()


for r in (0, 3, 7):
    own = [r, 2 * cp - 1 - r]
    pairs = c * (r + 0.5) * c + c * (2 * cp - 1 - r + 0.5) * c
    flops = 4.0 * D * Hq * pairs
    for n_split in (1, 2, 4):
        hg = Hkv // n_split
        rows = [torch.randn(cp * 2 * s_l, hg, D, generator=gen, device=DEV).bfloat16() for _ in range(n_split)]
        streams = [torch.cuda.Stream() for _ in range(n_split)]
        main = torch.cuda.current_stream()

        def launch(j):
            ops.flash_attn(q[:, :, j * hg * G:(j + 1) * hg * G], rows[j].unsqueeze(0), rows[j][s_l:].unsqueeze(0), causal=True, chunk_len=c,
                           q_chunk_gid=own, kv_chunk_gid=kv_gid, kv_chunk_row=kv_row, out=o[:, :, j * hg * G:(j + 1) * hg * G])

        def one_stream():
            for j in range(n_split):
                launch(j)

        def many_streams():
            ready = torch.cuda.Event(); ready.record(main)
            for j in range(n_split):
                st = main if j == 0 else streams[j]
                with torch.cuda.stream(st):
                    if st is not main:
                        st.wait_event(ready)
                    launch(j)
                    if st is not main:
                        ev = torch.cuda.Event(); ev.record(st); main.wait_event(ev)

        # the product's order (dot_product_attention.forward_cp): split 0 attends to the rank's OWN chunks first (while gather 0 would be in
        # flight), then to the 14 remote chunks, and merges; splits 1.. run on their own streams
        kv_own = torch.randn(2, s_l, hg, D, generator=gen, device=DEV).bfloat16()
        o_b = torch.empty(1, s_l, hg * G, D, dtype=torch.bfloat16, device=DEV)
        rem = [i for i in range(2 * cp) if i // 2 != r]

        def own_first(j):
            qj, oj = q[:, :, j * hg * G:(j + 1) * hg * G], o[:, :, j * hg * G:(j + 1) * hg * G]
            _, lse_a = ops.flash_attn(qj, kv_own[0].unsqueeze(0), kv_own[1].unsqueeze(0), causal=True, chunk_len=c, q_chunk_gid=own,
                                      kv_chunk_gid=own, kv_chunk_row=[0, c], out=oj, return_lse=True)
            _, lse_b = ops.flash_attn(qj, rows[j].unsqueeze(0), rows[j][s_l:].unsqueeze(0), causal=True, chunk_len=c, q_chunk_gid=own,
                                      kv_chunk_gid=[kv_gid[i] for i in rem], kv_chunk_row=[kv_row[i] for i in rem], out=o_b, return_lse=True)
            ops.attn_merge_(oj, lse_a, o_b, lse_b)

        def product_order():
            ready = torch.cuda.Event(); ready.record(main)
            evs = []
            for j in range(n_split):
                st = main if j == 0 else streams[j]
                with torch.cuda.stream(st):
                    if st is not main:
                        st.wait_event(ready)
                    own_first(j) if j == 0 else launch(j)
                    if st is not main:
                        ev = torch.cuda.Event(); ev.record(st); evs.append(ev)
            for ev in evs:
                main.wait_event(ev)

        variants = (("one stream", one_stream),) + ((("%d streams" % n_split, many_streams),) if n_split > 1 else ())
        variants += (("own chunks first for split 0, %d stream(s)" % n_split, product_order),)
        for name, fn in variants:
            fn(); torch.cuda.synchronize()
            ts = []
            for _ in range(5):
                a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                a.record(); fn(); b.record(); torch.cuda.synchronize()
                ts.append(a.elapsed_time(b))
            t = statistics.median(ts)
            emit(kind="cp8_rank_attn", rank=r, n_split=n_split, workgroups_per_launch=Hq * (s_l // 256) // n_split, how=name, ms=t,
                 tflops=flops / t / 1e9)
        del rows
