import os, sys, time, torch, torch.distributed as dist
sys.path.insert(0, ".")
dist.init_process_group("nccl", device_id=torch.device("cuda:0"))
from long_vita_amd import generation, gpt_vl_model, lib, parallel_state as mpu
lib.load(allow_build=False); mpu.initialize_model_parallel()
cfg = gpt_vl_model.GPTConfig()
seq = int(os.environ.get("SEQ", "16384"))
model = gpt_vl_model.GPTVLModel.random_init(cfg, seed=1234, device="cuda:0")
assert model.force_cp_path
tokens = torch.randint(0, 150000, (1, seq), device="cuda:0")
for streams in (0, 1, 0, 1):
    model.core_attention.split_streams = bool(streams)
    generation.prefill_step(model, tokens, seq, None, reference_compat=False); torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(3):
        out = generation.prefill_step(model, tokens, seq, None, reference_compat=False)
    torch.cuda.synchronize()
    print("forced CP, S_l = %d, split launches on %s: %.1f ms / prefill" % (seq, "4 streams" if streams else "1 stream", (time.perf_counter() - t0) / 3 * 1e3))
dist.destroy_process_group()
