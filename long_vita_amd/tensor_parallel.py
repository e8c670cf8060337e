"""Tensor-parallel sharding of the decoder (BASELINE config 5: TP = 2 x CP = 4) — the partitioning of
M/core/tensor_parallel/layers.py (ColumnParallelLinear: output rows, RowParallelLinear: input columns):

  linear_qkv  [ng (qpg+2) d, h]   rows of the rank's kv groups (the fused layout is per group, so a contiguous block)
  linear_proj [h, np d]           columns of the rank's heads
  linear_fc1  [2F, h] = [gate; up] the rank's rows of gate and of up, re-stacked
  linear_fc2  [h, F]              the rank's columns
  output_layer [V, h]             vocab-parallel rows
  norms, embedding                replicated (the reference also shards the embedding over the vocabulary; replicating it
                                  gives the same values and gradients and keeps the lookup a plain gather)

Row-parallel outputs are summed over the TP group in bf16 and the residual is added afterwards (vita_add_bf16), as
RowParallelLinear + bias_dropout_add do; column-parallel input gradients are summed the same way in the backward.
The frozen ViT is replicated (3.5 % of the prefill flops).  This file shards the STAND-ALONE step's parameter dict, which runs without
sequence parallelism; on the module path (layers.py: Column / RowParallelLinear with `sequence_parallel`, the embedding's scatter)
--sequence-parallel is built and tested.
"""
from __future__ import annotations

import dataclasses

import torch


def shard_llm_params(p: dict, cfg, tp_size: int, tp_rank: int):
    """Full Megatron-layout params -> (this rank's shard, its GPTConfig/LLMConfig with local heads / groups / ffn)."""
    if cfg.kv_groups % tp_size or cfg.ffn % tp_size or p["lm_head"].shape[0] % tp_size:
        raise ValueError("kv groups, ffn and vocabulary must divide by the tensor-parallel size")
    d, h = cfg.head_dim, cfg.hidden
    ng_l, f_l = cfg.kv_groups // tp_size, cfg.ffn // tp_size
    rows_qkv = ng_l * (cfg.qpg + 2) * d
    cols_o = (cfg.heads // tp_size) * d
    v_l = p["lm_head"].shape[0] // tp_size
    r = tp_rank
    out = {"embed": p["embed"], "final_ln": p["final_ln"], "lm_head": p["lm_head"][r * v_l: (r + 1) * v_l].contiguous(),
           "layers": []}
    for lp in p["layers"]:
        gate, up = lp["fc1_w"].chunk(2, dim=0)
        out["layers"].append({
            "ln1": lp["ln1"], "ln2": lp["ln2"],
            "qkv_w": lp["qkv_w"][r * rows_qkv: (r + 1) * rows_qkv].contiguous(),
            "qkv_b": lp["qkv_b"][r * rows_qkv: (r + 1) * rows_qkv].contiguous(),
            "o_w": lp["o_w"][:, r * cols_o: (r + 1) * cols_o].contiguous(),
            "fc1_w": torch.cat([gate[r * f_l: (r + 1) * f_l], up[r * f_l: (r + 1) * f_l]]).contiguous(),
            "fc2_w": lp["fc2_w"][:, r * f_l: (r + 1) * f_l].contiguous()})
    local_cfg = dataclasses.replace(cfg, heads=cfg.heads // tp_size, kv_groups=ng_l, ffn=f_l)
    return out, local_cfg


def unshard_llm_grads(shards, cfg, tp_size: int) -> dict:
    """Inverse of shard_llm_params for gradients (list over TP ranks -> full layout); replicated entries from rank 0."""
    full = {"embed": shards[0]["embed"], "final_ln": shards[0]["final_ln"],
            "lm_head": torch.cat([s["lm_head"] for s in shards], dim=0), "layers": []}
    for li in range(len(shards[0]["layers"])):
        ls = [s["layers"][li] for s in shards]
        halves = [l["fc1_w"].chunk(2, dim=0) for l in ls]
        full["layers"].append({
            "ln1": ls[0]["ln1"], "ln2": ls[0]["ln2"],
            "qkv_w": torch.cat([l["qkv_w"] for l in ls], dim=0), "qkv_b": torch.cat([l["qkv_b"] for l in ls], dim=0),
            "o_w": torch.cat([l["o_w"] for l in ls], dim=1),
            "fc1_w": torch.cat([hf[0] for hf in halves] + [hf[1] for hf in halves], dim=0),
            "fc2_w": torch.cat([l["fc2_w"] for l in ls], dim=1)})
    return full
