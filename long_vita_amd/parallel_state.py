"""Minimal context-parallel process-group state (the `mpu` accessors the hot path reads).

Mirrors the names of megatron.core.parallel_state that the reference calls on this path
(M/training/utils.py:276, M/core/models/common/embeddings/rotary_pos_embedding.py:37-38,
M/inference/text_generation/generation.py:518-519,543-544).  One process per GPU; the group is a
torch.distributed group (backend "nccl" = RCCL on ROCm, "gloo" in the CPU tests).
PP = DP = 1; TP = 1 on the prefill path (BASELINE configs 2-4), TP = 2 for the training step of config 5.
"""
from __future__ import annotations

from typing import Optional

import torch.distributed as dist

import threading


class _State(threading.local):
    """Thread-local so that the single-GPU tests can run several simulated ranks as threads."""
    group = None
    size = 1
    rank = 0
    tp_group = None
    tp_size = 1
    tp_rank = 0


_S = _State()


def initialize_model_parallel(context_parallel_size: Optional[int] = None, group=None,
                              tensor_model_parallel_size: int = 1) -> None:
    """Create the context-parallel group (and, for BASELINE config 5, the tensor-parallel groups) over the world.
    Megatron's rank order: tensor-parallel ranks are adjacent (rank = cp_rank * TP + tp_rank)."""
    if not dist.is_available() or not dist.is_initialized():
        _S.group, _S.size, _S.rank = None, 1, 0
        _S.tp_group, _S.tp_size, _S.tp_rank = None, 1, 0
        return
    world, me = dist.get_world_size(), dist.get_rank()
    tp = tensor_model_parallel_size
    cp = world // tp if context_parallel_size is None else context_parallel_size
    if cp * tp != world:
        raise ValueError("this path runs PP = DP = 1: context_parallel_size * tensor_model_parallel_size must equal the world size")
    if tp == 1:
        _S.group = group if group is not None else dist.group.WORLD
        _S.size, _S.rank = cp, me
        _S.tp_group, _S.tp_size, _S.tp_rank = None, 1, 0
        return
    # every rank creates every group, in the same order (torch.distributed.new_group is collective)
    for i in range(cp):
        ranks = list(range(i * tp, (i + 1) * tp))
        g = dist.new_group(ranks)
        if me in ranks:
            _S.tp_group, _S.tp_size, _S.tp_rank = g, tp, ranks.index(me)
    for j in range(tp):
        ranks = list(range(j, world, tp))
        g = dist.new_group(ranks)
        if me in ranks:
            _S.group, _S.size, _S.rank = g, cp, ranks.index(me)


_MEGATRON = None


def bind_megatron(mcore_parallel_state) -> None:
    """Under a real Megatron (megatron_adaptor.exe_adaptation) the accessors below answer from
    megatron.core.parallel_state whenever its model-parallel groups are initialised."""
    global _MEGATRON
    _MEGATRON = mcore_parallel_state


def _mg():
    m = _MEGATRON
    if m is not None and getattr(m, "model_parallel_is_initialized", lambda: False)():
        return m
    return None


def set_tensor_parallel_state(size: int, rank: int, group=None) -> None:
    _S.tp_group, _S.tp_size, _S.tp_rank = group, size, rank


def get_tensor_model_parallel_world_size() -> int:
    m = _mg()
    if m is not None:
        return m.get_tensor_model_parallel_world_size()
    return _S.tp_size


def get_tensor_model_parallel_rank() -> int:
    m = _mg()
    if m is not None:
        return m.get_tensor_model_parallel_rank()
    return _S.tp_rank


def get_tensor_model_parallel_src_rank() -> int:
    """Global rank of tensor-parallel rank 0 of this rank's group (Megatron: adjacent ranks form a TP group)."""
    m = _mg()
    if m is not None:
        return m.get_tensor_model_parallel_src_rank()
    import torch.distributed as dist
    if not dist.is_available() or not dist.is_initialized():
        return 0
    return (dist.get_rank() // _S.tp_size) * _S.tp_size


def get_tensor_model_parallel_group():
    m = _mg()
    if m is not None:
        return m.get_tensor_model_parallel_group()
    return _S.tp_group


def set_context_parallel_state(size: int, rank: int, group=None) -> None:
    """Explicit override (tests / embedding in a host framework that owns the groups)."""
    _S.group, _S.size, _S.rank = group, size, rank


def destroy_model_parallel() -> None:
    _S.group, _S.size, _S.rank = None, 1, 0
    _S.tp_group, _S.tp_size, _S.tp_rank = None, 1, 0


def get_context_parallel_world_size() -> int:
    m = _mg()
    if m is not None:
        return m.get_context_parallel_world_size()
    return _S.size


def get_context_parallel_rank() -> int:
    m = _mg()
    if m is not None:
        return m.get_context_parallel_rank()
    return _S.rank


def get_context_parallel_group():
    m = _mg()
    if m is not None:
        return m.get_context_parallel_group()
    return _S.group


def zigzag_chunk_ids(cp_size: int, cp_rank: int):
    """Chunks of the 2*CP-chunk sequence view owned by `cp_rank` (M/training/utils.py:329-341)."""
    return [cp_rank, 2 * cp_size - cp_rank - 1]
