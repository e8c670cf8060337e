"""Keep the core attention's output across Megatron's activation recompute (r05).

`--recompute-granularity full` re-runs a layer's whole forward in the backward (megatron/core/tensor_parallel/random.py:
CheckpointFunction — forward under no_grad keeping the inputs, backward = the function again with autograd on; the reference's
TransformerBlock calls it per layer of the recompute block, M/core/transformer/transformer_block.py:210-305).  At long context the
dominant part of that second forward is the attention itself (and, under context parallelism, its K / V all-gather), and its result
is fully determined by the first run: with 288 GB of HBM the context [s, heads x d] bf16 and the log-sum-exp [heads, s] fp32 of each
checkpointed layer (0.17 GB per 16K layer, 1.4 GB per 128K layer) can simply stay.

`checkpoint_wrapper` is registered by megatron_adaptor on `megatron.core.tensor_parallel.random.checkpoint`: it marks the FIRST run
of the checkpointed function as "store" and the run inside the backward as "replay".  HipDotProductAttention (the registered
`core_attention`) stores (context, lse) of every causal attention call of a store run and, in the replay, hands them to
FlashAttnFn / FlashAttnCPFn, which then skip the forward kernel (and the all-gather) and only save what their backward needs.  The
same kernel produced the kept tensors from the same inputs, so the loss and every gradient are bit-identical to recomputing.
Off unless VITA_KEEP_ATTENTION=1 (Megatron's flag means "keep the layer input only"); TrainStep's `keep_attention` is the same
idea on the stand-alone step (training.py)."""
from __future__ import annotations

import os
import threading
from functools import wraps

_S = threading.local()          # the backward of a checkpoint may run on autograd's thread: the phase is per thread, the region is shared


def enabled() -> bool:
    return os.environ.get("VITA_KEEP_ATTENTION", "0") not in ("", "0")


class Region:
    """One call of tensor_parallel.checkpoint: the attention results of its first run, in call order."""

    def __init__(self):
        self.slots, self.runs, self.cursor, self.broken = [], 0, 0, False


def current():
    """(region, phase) of the checkpointed function this thread is inside of, or (None, None)."""
    return getattr(_S, "region", None), getattr(_S, "phase", None)


def store(value) -> None:
    region, phase = current()
    if region is not None and phase == "store":
        region.slots.append(value)


def take(like=None):
    """The next kept result of the region being replayed, or None (nothing kept: the caller recomputes).
    EVERY attention call of the replay that stored in the store run must come here, whether or not it needs a gradient
    (HipDotProductAttention.forward does so): slots are matched to calls by call order, and a call that skipped its slot would hand
    the NEXT layer a result of the right shape and the wrong content.  `like` = the call's query [b, sq, np, hn]: a
    kept context of another shape means the two runs did not make the same calls — the region is then marked broken and this and
    every later call of it recomputes."""
    region, phase = current()
    if region is None or phase != "replay" or region.broken or region.cursor >= len(region.slots):
        return None
    value = region.slots[region.cursor]
    region.slots[region.cursor] = None          # the autograd node owns it from here on
    region.cursor += 1
    if like is not None and tuple(getattr(value[0], "shape", like.shape)) != tuple(like.shape):
        region.broken = True
        return None
    return value


def checkpoint_wrapper(fn):
    """Wrapper patch for `megatron.core.tensor_parallel.random.checkpoint(function, distribute_saved_activations, *args)`."""
    @wraps(fn)
    def wrapper(function, distribute_saved_activations, *args):
        if not enabled():
            return fn(function, distribute_saved_activations, *args)
        region = Region()

        def run(*a):
            prev = current()
            _S.region, _S.phase = region, ("store" if region.runs == 0 else "replay")
            region.runs += 1
            region.cursor = 0
            replay = region.runs > 1
            try:
                out = function(*a)
            finally:
                _S.region, _S.phase = prev
                left = len(region.slots) - region.cursor
                if replay:
                    region.slots.clear()
            if replay and left and not region.broken:
                # the replay made fewer attention calls than the store run kept results for: call order no longer identifies them
                raise RuntimeError(f"VITA_KEEP_ATTENTION: {left} kept attention result(s) were not consumed by the recompute of this "
                                   "checkpointed region; the store and replay runs made different calls")
            return out

        return fn(run, distribute_saved_activations, *args)

    return wrapper
