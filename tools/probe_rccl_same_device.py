"""Probe (VERDICT r04 item 1 iv): do TWO RCCL ranks run on the SAME MI355X?  A one-GPU box cannot hold a real world of two; if RCCL
accepted two communicator ranks on one device this would be the first world-2 execution of the exchange.  Writes what happened — the
all-gather's result or RCCL's refusal text — to gpurun_out/r05_rccl_same_device.txt (copied to profiles/).  Each rank runs under
`timeout`; nothing is killed by name.

    python tools/probe_rccl_same_device.py
"""
import os
import socket
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
WORKER = r"""
import os, torch, torch.distributed as dist
rank = int(os.environ["RANK"])
torch.cuda.set_device(0)
dist.init_process_group("nccl", device_id=torch.device("cuda:0"))
x = torch.full((1024,), float(rank + 1), device="cuda:0")
out = torch.empty(2048, device="cuda:0")
dist.all_gather_into_tensor(out, x)
torch.cuda.synchronize()
print("RESULT", rank, out[0].item(), out[1024].item(), flush=True)
dist.destroy_process_group()
"""


def main():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    procs = []
    for r in range(2):
        env = dict(os.environ, RANK=str(r), LOCAL_RANK="0", WORLD_SIZE="2", MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port),
                   HSA_ENABLE_IPC_MODE_LEGACY="0", NCCL_DEBUG="WARN")
        procs.append(subprocess.Popen(["timeout", "-k", "5", "150", sys.executable, "-c", WORKER], env=env, stdout=subprocess.PIPE,
                                      stderr=subprocess.STDOUT, text=True))
    text = []
    for r, p in enumerate(procs):
        out, _ = p.communicate()
        text.append(f"---- rank {r}: exit code {p.returncode} ----\n{out[-3000:]}")
    os.makedirs(os.path.join(ROOT, "gpurun_out"), exist_ok=True)
    path = os.path.join(ROOT, "gpurun_out", "r05_rccl_same_device.txt")
    import torch
    head = (f"two torch.distributed 'nccl' (RCCL) ranks, both on cuda:0 of a one-GPU MI355X box; torch {torch.__version__}\n"
            f"all_gather_into_tensor of 1024 floats per rank; each rank under `timeout 150`\n")
    open(path, "w").write(head + "\n".join(text) + "\n")
    print(head + "\n".join(text))


if __name__ == "__main__":
    main()
