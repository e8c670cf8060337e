"""Seeded synthetic prefill requests in the reference's token / index layout
(M/tasks/inference/module.py:640-700: per frame `<vid>` + 256 x `<VID_CONTEXT>` + `</vid>`,
`indices` [2, N, 256] = (batch 0, global positions of the context tokens); sequence padded to a
multiple of 64, :687).  Sizes follow SURVEY.md §8d."""
from __future__ import annotations

import torch

VID_START_ID, VID_CONTEXT_ID, VID_END_ID = 151665, 151666, 151667      # placeholders inside the 152064 vocab
TEXT_VOCAB = 151643
TOKENS_PER_FRAME = 256


def frames_for_seq(seq_len: int, tail_text: int = 64) -> int:
    """Largest frame count whose 258-token blocks + `tail_text` text tokens fit seq_len."""
    return max((seq_len - tail_text) // (TOKENS_PER_FRAME + 2), 0)


def make_request(seq_len: int, n_frames: int, image_size: int = 448, seed: int = 1234, device="cuda",
                 images: bool = True):
    """-> tokens [1, seq_len] int64, external_inputs {"images" [N,3,H,W] bf16, "indices" [2,N,256]} (or None)."""
    if seq_len % 64:
        raise ValueError("seq_len must be a multiple of 64 (module.py:687)")
    if n_frames * (TOKENS_PER_FRAME + 2) > seq_len:
        raise ValueError("frames do not fit")
    g = torch.Generator().manual_seed(seed)
    ids, pos = [], []
    for _ in range(n_frames):
        ids.append(VID_START_ID)
        pos.append(torch.arange(len(ids), len(ids) + TOKENS_PER_FRAME))
        ids += [VID_CONTEXT_ID] * TOKENS_PER_FRAME
        ids.append(VID_END_ID)
    n_text = seq_len - len(ids)
    tokens = torch.cat([torch.tensor(ids, dtype=torch.int64),
                        torch.randint(0, TEXT_VOCAB, (n_text,), generator=g)])[None]
    if n_frames == 0:
        return tokens.to(device), None
    indices = torch.stack([torch.zeros(n_frames, TOKENS_PER_FRAME, dtype=torch.int64), torch.stack(pos)])
    ext = {"indices": indices.to(device)}
    if images:
        gd = torch.Generator(device=device).manual_seed(seed + 1)
        ext["images"] = torch.randn(n_frames, 3, image_size, image_size, generator=gd, device=device).to(torch.bfloat16)
    return tokens.to(device), ext
