"""Secondary measurement: per-token decode latency of the 14B decoder against a
KV cache of `--context` rows on one GPU, beside the reference behaviour under CP (re-prefill per token =
bench.py's prefill time).  The cache is filled with synthetic rows (its content does not change the timing).
    python tools/bench_decode.py --context 131072 [--layers 48] [--tokens 16]
Prints one JSON line (not the contract line of bench.py): HBM-bound roofline = bytes of weights + cache rows
streamed per token / time."""
import argparse, json, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from long_vita_amd import gpt_vl_model, inference_params, lib, ops

ap = argparse.ArgumentParser()
ap.add_argument("--context", type=int, default=131072)
ap.add_argument("--layers", type=int, default=48)
ap.add_argument("--tokens", type=int, default=16)
args = ap.parse_args()
lib.load(allow_build=False)
dev = "cuda:0"
cfg = gpt_vl_model.GPTConfig(num_layers=args.layers)
model = gpt_vl_model.GPTVLModel.random_init(cfg, seed=1, device=dev)
ip = inference_params.InferenceParams(1, args.context + args.tokens + 8)
cap = args.context + args.tokens + 8
buf = (torch.randn(cfg.num_layers, 2, cap, cfg.kv_groups, cfg.head_dim, device=dev, dtype=torch.bfloat16))
ip.key_value_memory_dict = {li + 1: buf[li] for li in range(cfg.num_layers)}
ip.local_len = args.context
ip.sequence_len_offset = args.context
tok = torch.tensor([[17]], device=dev)


def step():
    pos = torch.tensor([[ip.sequence_len_offset]], device=dev)
    out = model(tok, pos, None, inference_params=ip)
    ip.sequence_len_offset += 1
    return out


for _ in range(3):
    step()
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(args.tokens):
    out = step()
torch.cuda.synchronize()
dt = (time.perf_counter() - t0) / args.tokens
assert torch.isfinite(out.float()).all()
w_bytes = cfg.num_layers * 2 * (cfg.hidden * cfg.qkv_out + cfg.hidden * cfg.heads * cfg.head_dim + 3 * cfg.hidden * cfg.ffn)
w_bytes += 2 * cfg.vocab * cfg.hidden                                   # LM head
kv_bytes = cfg.num_layers * 2 * args.context * cfg.kv_groups * cfg.head_dim * 2
# kernel-only time of the two dominant kernels
ev = [torch.cuda.Event(enable_timing=True) for _ in range(4)]
lp = model.p["layers"][0]
x = torch.randn(cfg.hidden, device=dev).to(torch.bfloat16)
act = torch.empty(cfg.ffn, device=dev, dtype=torch.bfloat16)
q = torch.randn(cfg.kv_groups, cfg.qpg, cfg.head_dim, device=dev).to(torch.bfloat16)
nl = cfg.num_layers
ev[0].record()
for i in range(96):                    # rotate over the layers' weights: 48 x 141 MB does not fit the 256 MB MALL
    ops.gemv(x, model.p["layers"][i % nl]["fc1_w"], ops.EPI_SWIGLU, out=act)
ev[1].record()
ev[2].record()
for i in range(96):
    pm, pl, po = ops.decode_attn_partial(q, buf[i % nl, 0], buf[i % nl, 1], args.context)
ev[3].record()
torch.cuda.synchronize()
t_fc1 = ev[0].elapsed_time(ev[1]) / 96
t_att = ev[2].elapsed_time(ev[3]) / 96
print(json.dumps({"what": "decode, 1 token vs sharded KV cache, CP=1", "context": args.context, "layers": cfg.num_layers,
                  "ms_per_token": dt * 1e3, "tokens_per_s": 1 / dt,
                  "hbm_bytes_per_token_GB": (w_bytes + kv_bytes) / 1e9,
                  "achieved_GBps_end_to_end": (w_bytes + kv_bytes) / dt / 1e9,
                  "fc1_gemv_ms": t_fc1, "fc1_gemv_GBps": 2 * 2 * cfg.ffn * cfg.hidden / t_fc1 / 1e6,
                  "decode_attn_ms": t_att,
                  "decode_attn_GBps": 2 * args.context * cfg.kv_groups * cfg.head_dim * 2 / t_att / 1e6}))
