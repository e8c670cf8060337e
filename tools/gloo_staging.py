"""DIAGNOSTIC transport: torch.distributed collectives on device tensors carried by a gloo process group through host memory.

A one-GPU box cannot hold a real RCCL world of two (RCCL refuses two ranks on one device: profiles/r05_rccl_same_device.txt), so the
N > 1 code of bench.py — rank launch, process-group set-up, the zig-zag split of the request, K / V all-gather per layer, logits
gather, the first-step vote, `comm`, `cross_rank_check` — had only ever run on a world of one.  With `VITA_BENCH_BACKEND=gloo-staged`
(accepted together with `--dry-run` only: the timing means nothing) bench.py initialises a gloo group, maps every rank to the same
device, and `install()` wraps the four collectives the path uses so that a device tensor is copied to the host, exchanged between
the PROCESSES by gloo, and copied back.  Everything else — every kernel, every chunk table, every rank-dependent index — is the
product's own code running in N separate processes.  Never a measurement, never on the product path."""
import torch
import torch.distributed as dist


class _Done:
    def wait(self, *a, **k):
        return True

    def is_completed(self):
        return True


def install():
    real = {n: getattr(dist, n) for n in ("all_gather_into_tensor", "all_reduce", "reduce_scatter_tensor", "broadcast")}

    def staged(name, out_arg, in_args):
        fn = real[name]

        def wrapper(*args, async_op=False, **kw):
            args = list(args)
            dev = [a for a in args if torch.is_tensor(a) and a.is_cuda]
            if not dev:
                return fn(*args, async_op=async_op, **kw)
            torch.cuda.current_stream().synchronize()            # the producers of the send buffer have finished
            host = [a.detach().cpu() if torch.is_tensor(a) and a.is_cuda else a for a in args]
            if host[out_arg].dtype == torch.bfloat16 and name in ("all_reduce", "reduce_scatter_tensor"):
                # gloo sums bf16 through fp32 here; RCCL sums in bf16 — a diagnostic transport, parity limits account for neither
                host = [h.float() if torch.is_tensor(h) else h for h in host]
            fn(*host, async_op=False, **kw)
            args[out_arg].copy_(host[out_arg].to(args[out_arg].dtype))
            return _Done() if async_op else None

        return wrapper

    dist.all_gather_into_tensor = staged("all_gather_into_tensor", 0, (1,))
    dist.all_reduce = staged("all_reduce", 0, (0,))
    dist.reduce_scatter_tensor = staged("reduce_scatter_tensor", 0, (1,))
    dist.broadcast = staged("broadcast", 0, (0,))
