"""Zig-zag context-parallel batch slice — device-side mirror of M/training/utils.py.

`get_batch_on_this_cp_rank` keeps the reference's name, argument meaning and key behaviour
(M/training/utils.py:252-343; inference twin M/inference/text_generation/generation.py:517-539):
  * every [b, s, ...] tensor is viewed as 2*CP chunks and the rank keeps chunks {r, 2CP-1-r};
  * `external_images` keeps only frames with at least one token on this rank;
  * `external_indices` [2, N, L] is replaced by `external_src_indices` (row in the *selected*
    frames, token column) and `external_tgt_indices` (batch, *local* position).
The index arithmetic runs in HIP kernels (csrc/rows.hip); the reference's
`torch.isin` + `index_of_a_in_b` (:347-350) is replaced by the closed form
chunk = pos // C, local = (chunk == r ? 0 : C) + pos % C, which is bit-identical for the unique
positions the reference requires.
"""
from __future__ import annotations

from typing import Optional

import torch

from . import ops, parallel_state as mpu

_ACTUAL_SEQ_LEN = None
_POSITION_IDS = None


def get_actual_seq_len():
    return _ACTUAL_SEQ_LEN


def set_actual_seq_len(actual_seq_len):
    global _ACTUAL_SEQ_LEN
    _ACTUAL_SEQ_LEN = actual_seq_len


def get_position_ids():
    return _POSITION_IDS


def set_position_ids(position_ids):
    global _POSITION_IDS
    _POSITION_IDS = position_ids


def zigzag_slice(val: torch.Tensor, cp_size: int, cp_rank: int, seq_dim: int = 1) -> torch.Tensor:
    """val[..., s, ...] -> the rank's two chunks, concatenated (utils.py:329-341)."""
    s = val.shape[seq_dim]
    if s % (2 * cp_size):
        raise ValueError(f"sequence length {s} not divisible by 2*CP = {2 * cp_size}")
    c = s // (2 * cp_size)
    a, b = mpu.zigzag_chunk_ids(cp_size, cp_rank)
    return torch.cat([val.narrow(seq_dim, a * c, c), val.narrow(seq_dim, b * c, c)], dim=seq_dim)


def get_batch_on_this_cp_rank(batch: dict, seq_length: Optional[int] = None, cp_size: Optional[int] = None,
                              cp_rank: Optional[int] = None, reset_position_ids: bool = False) -> dict:
    cp_size = mpu.get_context_parallel_world_size() if cp_size is None else cp_size
    cp_rank = mpu.get_context_parallel_rank() if cp_rank is None else cp_rank
    if reset_position_ids:
        set_position_ids(batch["position_ids"].transpose(0, 1).contiguous())      # utils.py:267-270
    if cp_size <= 1:
        return batch
    batch = dict(batch)
    if seq_length is None:
        seq_length = next(v.shape[1] for k, v in batch.items()
                          if v is not None and not k.startswith("external_") and k != "attention_mask")

    hit = local = selected = None
    if "external_indices" in batch:
        ind = batch["external_indices"]
        indices_s = ind[1].contiguous()
        hit, local = ops.cp_index_remap(indices_s, seq_length, cp_size, cp_rank)    # [N, L] uint8 / int64
        any_hit = bool(hit.any().item())
        if any_hit:
            selected = ops.mask_to_index(ops.rows_any(hit))                          # frames on this rank

    for key, val in list(batch.items()):
        if key == "external_images":
            if selected is not None:
                batch[key] = ops.row_gather(val.contiguous(), selected)
            continue
        if key == "external_indices":
            if selected is not None:
                n_img, tok = hit.shape
                hit_idx = ops.mask_to_index(hit)
                img_rank = ops.index_inverse(selected, n_img)
                sb, ss, tb, ts = ops.cp_src_tgt(hit_idx, tok, img_rank, val[0].reshape(-1), local.reshape(-1))
                batch["external_src_indices"] = torch.stack([sb, ss])
                batch["external_tgt_indices"] = torch.stack([tb, ts])
            batch.pop(key)
            continue
        if key == "attention_mask" or val is None:
            continue
        batch[key] = zigzag_slice(val, cp_size, cp_rank, seq_dim=1)
    return batch


def get_batch_on_this_tp_rank(data_iterator, *, micro_batch_size: int, seq_length: int, image_size: int = 448,
                              reset_attention_mask: bool = False, create_attention_mask_in_dataloader: bool = False,
                              device=None) -> dict:
    """Mirror of M/training/utils.py:410-626 at pipeline size 1 (the Long-VITA path runs PP = 1): tensor-parallel rank 0
    draws the batch (skipping items without "tokens", :434-438), casts the frames to bf16 (:450; a batch without images gets the
    reference's all-ones placeholder frame per sample, :451-452) and broadcasts, in the reference's ORDER, tokens / labels /
    loss_mask / attention_mask / position_ids, the two size vectors, the frames and — only when present — the context-token
    indices over the tensor-parallel group; the other ranks allocate by `micro_batch_size` / `seq_length` and the received
    sizes (:525-545).  `--reset-attention-mask`: `actual_seq_len` goes through the length-prefixed dynamic broadcast
    (:419-431,513-516) into set_actual_seq_len.  Pure data movement: torch.distributed over RCCL on device tensors
    (gloo + host tensors in the CPU tests); no arithmetic."""
    import torch.distributed as dist
    tp, rank, group = mpu.get_tensor_model_parallel_world_size(), mpu.get_tensor_model_parallel_rank(), mpu.get_tensor_model_parallel_group()
    if device is None:
        device = torch.device("cuda", torch.cuda.current_device()) if torch.cuda.is_available() else torch.device("cpu")
    src = mpu.get_tensor_model_parallel_src_rank()

    def _broadcast(item):
        if item is not None and tp > 1:
            dist.broadcast(item, src, group=group)

    def broadcast_dynamic(item):
        if item is not None:
            item = item.to(device)
            item_len = torch.tensor(item.numel(), device=device)
            _broadcast(item_len)
            _broadcast(item)
        else:
            item_len = torch.empty((), dtype=torch.int64, device=device)
            _broadcast(item_len)
            item = torch.empty([int(item_len.item())], dtype=torch.int64, device=device)
            _broadcast(item)
        return item

    if rank == 0:
        data = None
        if data_iterator is not None:
            while True:
                data = next(data_iterator)
                if "tokens" in data:
                    break
        to = lambda t: t.to(device, non_blocking=True)                                             # noqa: E731
        batch = {"tokens": to(data["tokens"]), "labels": to(data["labels"]), "loss_mask": to(data["loss_mask"]),
                 "attention_mask": None if "attention_mask" not in data else to(data["attention_mask"]),
                 "position_ids": to(data["position_ids"])}
        if "images" in data:
            batch["external_images"] = to(data["images"]).bfloat16()
        else:
            batch["external_images"] = torch.ones([len(data["tokens"]), 3, image_size, image_size]).to(device).bfloat16()
        external_images_sizes = torch.tensor(batch["external_images"].size()).to(device)
        if "image_indices" in data:
            batch["external_indices"] = to(data["image_indices"]).to(torch.int64)
            external_indices_sizes = torch.tensor(batch["external_indices"].size()).to(device)
        else:
            external_indices_sizes = torch.tensor([0, 0, 0]).to(device)
        for k in ("tokens", "labels", "loss_mask", "attention_mask", "position_ids"):
            _broadcast(batch[k])
        _broadcast(external_images_sizes)
        _broadcast(external_indices_sizes)
        _broadcast(batch["external_images"])
        if external_indices_sizes.sum() > 0:
            _broadcast(batch["external_indices"])
        if reset_attention_mask:
            set_actual_seq_len(broadcast_dynamic(data["actual_seq_len"]).tolist())
        return batch

    tokens = torch.empty((micro_batch_size, seq_length), dtype=torch.int64, device=device)
    labels = torch.empty((micro_batch_size, seq_length), dtype=torch.int64, device=device)
    loss_mask = torch.empty((micro_batch_size, seq_length), dtype=torch.float32, device=device)
    attention_mask = (torch.empty((micro_batch_size, 1, seq_length, seq_length), dtype=torch.bool, device=device)
                      if create_attention_mask_in_dataloader else None)
    position_ids = torch.empty((micro_batch_size, seq_length), dtype=torch.int64, device=device)
    external_images_sizes = torch.empty((4), dtype=torch.int64, device=device)
    external_indices_sizes = torch.empty((3), dtype=torch.int64, device=device)
    for t in (tokens, labels, loss_mask, attention_mask, position_ids, external_images_sizes, external_indices_sizes):
        _broadcast(t)
    external_images = torch.empty(external_images_sizes.tolist(), dtype=torch.bfloat16, device=device)
    _broadcast(external_images)
    external_indices = None
    if external_indices_sizes.sum() > 0:
        external_indices = torch.empty(external_indices_sizes.tolist(), dtype=torch.int64, device=device)
        _broadcast(external_indices)
    if reset_attention_mask:
        set_actual_seq_len(broadcast_dynamic(None).tolist())
    batch = {"tokens": tokens, "labels": labels, "loss_mask": loss_mask, "attention_mask": attention_mask,
             "position_ids": position_ids, "external_images": external_images}
    if external_indices is not None:
        batch["external_indices"] = external_indices
    return batch


def get_packed_segments():
    """Packed samples as the reference's GPU path detects them: transformers' _flash_attention_forward
    (called at M/core/transformer/dot_product_attention.py:374-390 with position_ids=get_position_ids()) switches to
    flash_attn_varlen when the position ids are not monotonic, with cu_seqlens at the zeros of position_ids.
    Returns (seg_start, seg_end) int32 [s] on the device, or None.  CP = 1 only."""
    pid = _POSITION_IDS
    if pid is None:
        return None
    cached = getattr(get_packed_segments, "_cache", None)
    if cached is not None and cached[0] is pid:
        return cached[1]
    pos = pid.reshape(pid.shape[0], -1)[:, 0]                              # stored [s, b] (:268-270), b == 1
    seg = None
    if pos.numel() > 1 and not bool((torch.diff(pos) >= 0).all()):
        from . import ops
        cu = (pos == 0).nonzero().flatten()
        if cu.numel() == 0 or int(cu[0]) != 0:
            cu = torch.cat([cu.new_zeros(1), cu])
        seg = ops.segments_from_cu_seqlens(cu, pos.numel())
    get_packed_segments._cache = (pid, seg)
    return seg
