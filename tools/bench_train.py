"""Secondary measurement (BASELINE config 5 shape at TP=1): forward + backward step
time of the 14B decoder with the logits-masked head and full activation recompute, one GPU.
    python tools/bench_train.py --seq 16384 [--layers 48]
Prints one JSON line (not the contract line of bench.py)."""
import argparse, json, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from long_vita_amd import gpt_vl_model, lib, training

ap = argparse.ArgumentParser()
ap.add_argument("--seq", type=int, default=16384)
ap.add_argument("--layers", type=int, default=48)
ap.add_argument("--answer", type=int, default=512)
ap.add_argument("--steps", type=int, default=2)
ap.add_argument("--recompute-num-layers", type=int, default=-1, help="-1 = every layer (full recompute); stage 3 passes 20; 0 keeps all activations")
ap.add_argument("--keep-attention", type=int, default=0, help="1: the recompute block keeps its attention half (TrainStep keep_attention)")
args = ap.parse_args()
lib.load(allow_build=False)
dev = "cuda:0"
cfg = gpt_vl_model.GPTConfig(num_layers=args.layers)
model = gpt_vl_model.GPTVLModel.random_init(cfg, seed=1, device=dev)
g = torch.Generator(device=dev).manual_seed(2)
S = args.seq
tokens = torch.randint(0, 151643, (1, S), generator=g, device=dev)
labels = torch.roll(tokens, -1, 1)
loss_mask = torch.zeros(1, S, device=dev)
loss_mask[0, S - args.answer:] = 1
rec = None if args.recompute_num_layers < 0 else args.recompute_num_layers
step = training.TrainStep(model, recompute_num_layers=rec, keep_attention=bool(args.keep_attention))
loss, grads = step.forward_backward(tokens, labels, loss_mask)      # warm-up
del grads
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(args.steps):
    loss, grads = step.forward_backward(tokens, labels, loss_mask)
    del grads
torch.cuda.synchronize()
dt = (time.perf_counter() - t0) / args.steps
lin = cfg.num_layers * 2 * (cfg.hidden * cfg.qkv_out + cfg.hidden * cfg.heads * cfg.head_dim + 3 * cfg.hidden * cfg.ffn) * S
attn = cfg.num_layers * 4 * cfg.head_dim * cfg.heads * (S * (S + 1) // 2)
fwd = lin + attn
# fwd + bwd (2x linears; attention backward = 5 of the forward's 2 GEMM units = 2.5x) + the recompute forward of the layers in the
# recompute block ("3x forward FLOPs + recompute factor as configured")
n_rec = cfg.num_layers if rec is None else max(0, min(cfg.num_layers, rec))
r = n_rec / cfg.num_layers
alg = (3 + r) * lin + (3.5 + r) * attn
print(json.dumps({"what": f"train step fwd+bwd, recompute block = {n_rec} of {cfg.num_layers} layers, TP=1 CP=1" + (", attention half kept (keep_attention)" if args.keep_attention else ""), "keep_attention": bool(args.keep_attention), "seq": S,
                  "layers": cfg.num_layers, "recompute_num_layers": n_rec,
                  "answer_tokens": args.answer, "s_per_step": dt, "loss": float(loss),
                  "algorithmic_tflop_per_step": alg / 1e12, "tflops": alg / dt / 1e12,
                  "tflops_without_recompute_work": (3 * lin + 3.5 * attn) / dt / 1e12,
                  "tokens_per_s": S / dt, "peak_mem_gb": torch.cuda.max_memory_allocated() / 2**30}))
