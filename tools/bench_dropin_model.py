"""The WHOLE training step the reference's entry script drives, at BASELINE config 2's sequence on one GPU: forward_step -> GPTVLModel.forward
with labels + logit mask -> loss_func -> loss.backward() through autograd over the registered modules (48 full-width decoder layers,
vocabulary 152064, 16384 tokens, 512 answer tokens, text only) — `tests/dummy_megatron.py` restates GPTVLModel / TransformerBlock /
tensor_parallel.checkpoint / forward_step / loss_func (Megatron-LM is not installable here) — next to training.TrainStep's explicit
sweep on the same sizes (tools/bench_train.py: profiles/r05_train_step_16k_n1.jsonl).

    python tools/bench_dropin_model.py [recompute_num_layers ...]      (0 = every activation kept, 20 = stage 3's flag, 48 = all)
Appends JSON lines to gpurun_out/r05_dropin_model.jsonl."""
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch  # noqa: E402

import dummy_megatron as dm  # noqa: E402
from long_vita_amd import lib  # noqa: E402
from long_vita_amd.patch_utils import MindSpeedPatchesManager as aspm  # noqa: E402

lib.load(allow_build=False)
dm.install()
import long_vita_amd.megatron_adaptor as ad  # noqa: E402
aspm.patches_info = {}
assert ad.exe_adaptation(create_dummy=True)
specs = sys.modules["megatron.core.models.gpt.gpt_layer_specs"]
gpt_cls = sys.modules["long_vita_megatron.core.models.multimodal.gpt_vl_model"].GPTVLModel
DEV = "cuda"
os.makedirs(os.path.join(ROOT, "gpurun_out"), exist_ok=True)
LOG = open(os.path.join(ROOT, "gpurun_out", "r05_dropin_model.jsonl"), "a")
S, V, L, ANSWER = 16384, 152064, 48, 512


def run(n_rec):
    mcfg = dm.TransformerConfig(num_layers=L, hidden_size=5120, num_attention_heads=40, num_query_groups=8, kv_channels=128,
                                ffn_hidden_size=13824, recompute_granularity="full" if n_rec else None,
                                recompute_method="block" if n_rec else None, recompute_num_layers=n_rec or None)
    torch.manual_seed(1)
    model = gpt_cls(mcfg, transformer_layer_spec=specs.get_gpt_layer_with_transformer_engine_spec(), vocab_size=V, max_sequence_length=S,
                    position_embedding_type="rope", rotary_base=1e6, external_feature_model_provider=lambda cfg: None)
    model.unused.data = model.unused.data.to(DEV).bfloat16()
    model.train()
    n_params = sum(q.numel() for q in model.parameters())
    g = torch.Generator(device=DEV).manual_seed(2)
    tokens = torch.randint(0, 151643, (1, S), generator=g, device=DEV)
    labels = torch.roll(tokens, -1, 1)
    loss_mask = torch.zeros(1, S, device=DEV)
    loss_mask[0, S - ANSWER:] = 1
    position_ids = torch.arange(S, dtype=torch.long, device=DEV).unsqueeze(0)
    batch = (tokens, labels, loss_mask, None, position_ids, {})

    def step():
        for q in model.parameters():
            q.grad = None
        out, lf = dm.forward_step(batch, model)
        loss_sum, n_tok = lf(out)
        loss = loss_sum / n_tok
        loss.backward()
        return loss

    loss = step()                                            # warm-up (allocator, RoPE tables)
    torch.cuda.synchronize()
    for q in model.parameters():
        q.grad = None
    torch.cuda.empty_cache()
    base = torch.cuda.memory_allocated()
    torch.cuda.reset_peak_memory_stats()
    out, lf = dm.forward_step(batch, model)
    torch.cuda.synchronize()
    kept = torch.cuda.memory_allocated() - base
    loss_sum, n_tok = lf(out)
    (loss_sum / n_tok).backward()
    torch.cuda.synchronize()
    peak = torch.cuda.max_memory_allocated()
    ts = []
    for _ in range(2):
        torch.cuda.synchronize(); t0 = time.perf_counter()
        loss = step()
        torch.cuda.synchronize(); ts.append(time.perf_counter() - t0)
    rec = dict(kind="dropin_model_step", what="forward_step -> GPTVLModel.forward -> loss_func -> backward through autograd over the registered "
               "modules (tests/dummy_megatron.py stands in for Megatron-LM)", seq=S, layers=L, vocab=V, answer_tokens=ANSWER, parameters=n_params,
               recompute_num_layers=n_rec, VITA_KEEP_ATTENTION=os.environ.get("VITA_KEEP_ATTENTION", "0"), s_per_step=min(ts), s_per_step_all=ts, loss=float(loss),
               activations_kept_after_forward_gb=kept / 1e9, peak_allocated_gb=peak / 1e9, weights_gb=base / 1e9)
    s = json.dumps(rec)
    print(s, flush=True)
    LOG.write(s + "\n"); LOG.flush()
    del model


for a in (sys.argv[1:] or ["0", "20", "48"]):
    run(int(a))
    torch.cuda.empty_cache()
