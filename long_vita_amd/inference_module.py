"""Request -> tensors: mirror of get_external_inputs (M/tasks/inference/module.py:493-707), the step in front of the
prefill.  Token-id surgery is host-side integer work and follows the reference statement by
statement (bit-exact: fixture made by the reference's own function, tests/golden/external_inputs.pt):

  * every `<image>` tag becomes  <img> + 256 x <IMG_CONTEXT> + </img>  and, when the image was tiled
    (process_dynamic returned more than one patch), one  "\\n" + per tile <patch> + 256 x <PATCH_CONTEXT> + </patch>
    row per tile row (:551-614);
  * every `<video>` tag becomes, per frame,  <vid> + 256 x <VID_CONTEXT> + </vid>  (:619-678);
  * `indices` [2, n_images, 256] records (batch, position) of every context token in order (:575-585);
  * each row is padded to a multiple of 64 with pad (or eos) (:684-686).

The pixels come from long_vita_amd.image_processor.ImageProcessor (GPU kernels) instead of PIL on the host; get_args()
values are keyword arguments.  `video_frames_list` (already decoded frames, one [N, H, W, 3] uint8 array per <video> tag)
is this framework's addition: file decoding (decord) is out of scope.

Per-rank loading (cp_size / cp_rank given).  The reference builds all N frames on rank 0 and broadcasts them to the
world (4.9 GB at 4096 frames, module.py:340-360); get_batch_on_this_cp_rank (generation.py:517-539) then throws away, on
every rank, the frames without a token in the rank's two zig-zag chunks.  Which frames those are depends only on the
token layout, so here every rank lays the tokens out first (no pixel touched), keeps the frames
`frames_on_this_cp_rank` names, and decodes / uploads / resizes only those: the returned images / indices are exactly the
rows the reference's selection would keep, and that selection is idempotent on them (every returned frame has a token
on the rank; source rows are renumbered over the selected frames), so everything downstream is unchanged."""
from __future__ import annotations

from typing import Optional, Sequence

import torch

# long_vita/constants.py:5-22 (the active `if True:` set)
IMG_TAG_TOKEN, VID_TAG_TOKEN = "<image>", "<video>"
IMG_CONTEXT_TOKEN, IMG_START_TOKEN, IMG_END_TOKEN = "<IMG_CONTEXT>", "<img>", "</img>"
VID_CONTEXT_TOKEN, VID_START_TOKEN, VID_END_TOKEN = "<VID_CONTEXT>", "<vid>", "</vid>"
PATCH_CONTEXT_TOKEN, PATCH_START_TOKEN, PATCH_END_TOKEN = "<PATCH_CONTEXT>", "<patch>", "</patch>"


def _single_id(tokenizer, text: str) -> int:
    ids = tokenizer(text, add_special_tokens=False).input_ids
    assert len(ids) == 1, f"{text!r} must be one token"                       # :525-535
    return ids[0]


def frames_on_this_cp_rank(starts: Sequence[int], length: int, seq_length: int, cp_size: int, cp_rank: int):
    """keep[i] = does the context-token run [starts[i], starts[i] + length) intersect the rank's zig-zag chunks
    {r, 2CP-1-r} of a sequence of seq_length tokens — the frame selection of M/training/utils.py:279-289
    (mask = isin(indices_s, calibration_index); image kept iff mask.any(dim=-1)) in closed form."""
    if seq_length % (2 * cp_size):
        raise ValueError(f"sequence length {seq_length} not divisible by 2*CP = {2 * cp_size}")
    c = seq_length // (2 * cp_size)
    owned = ((cp_rank * c, (cp_rank + 1) * c), ((2 * cp_size - 1 - cp_rank) * c, (2 * cp_size - cp_rank) * c))
    return [any(st < hi and st + length > lo for lo, hi in owned) for st in starts]


def _video_sources(image_processor, vid_idx, image_list, image_path_list, video_path_list, video_frames_list,
                   max_num_frame, max_fps):
    """The frames a <video> tag stands for, as things process_images accepts, without touching a pixel
    (same precedence as the eager branch below: the last list given wins)."""
    import os
    src = None
    if video_path_list is not None:
        path = video_path_list[vid_idx]
        if not os.path.isdir(path):
            raise NotImplementedError("video files need a decoder (decord is not part of this framework): "
                                      "decode the frames ImageProcessor.video_frame_indices names and pass video_frames_list")
        src = image_processor.directory_frame_paths(path, max_num_frame, max_fps)
    if image_path_list is not None:
        src = [image_path_list[vid_idx]]
    if image_list is not None:
        src = [image_list[vid_idx]]
    if video_frames_list is not None:
        src = list(video_frames_list[vid_idx])[:max_num_frame]
    return src


def _open_frames(sources):
    from PIL import Image
    return [Image.open(x).convert("RGB") if isinstance(x, str) else x for x in sources]


def get_external_inputs(tokens, image_list, image_path_list, video_path_list, tokenizer, image_processor, *,
                        image_token_length: int = 256, max_num_frame: int = 4096, max_fps: int = 1, bf16: bool = True,
                        video_frames_list: Optional[Sequence] = None, device="cuda",
                        cp_size: Optional[int] = None, cp_rank: Optional[int] = None, cp_seq_length: Optional[int] = None):
    """cp_size / cp_rank: per-rank loading (module docstring); cp_seq_length = the length the sequence is chunked at when
    that is not the 64-padded row (generation._cp_prefill_length pads a KV-cache prefill to 2*CP*256) — an int, or a
    callable of the expanded row's token count when it depends on it."""
    per_rank = cp_size is not None and cp_size > 1
    if per_rank and (cp_rank is None or not 0 <= cp_rank < cp_size):
        raise ValueError("per-rank loading needs 0 <= cp_rank < cp_size")
    tokens = tokens.tolist()
    if per_rank and len(tokens) != 1:
        raise ValueError("per-rank loading runs batch 1")
    pending = {}                                                               # slot in `images` -> frame sources
    IMG_CONTEXT_ID, IMG_START_ID, IMG_END_ID = (_single_id(tokenizer, t) for t in (IMG_CONTEXT_TOKEN, IMG_START_TOKEN, IMG_END_TOKEN))
    VID_CONTEXT_ID, VID_START_ID, VID_END_ID = (_single_id(tokenizer, t) for t in (VID_CONTEXT_TOKEN, VID_START_TOKEN, VID_END_TOKEN))
    PATCH_CONTEXT_ID, PATCH_START_ID, PATCH_END_ID = (_single_id(tokenizer, t) for t in (PATCH_CONTEXT_TOKEN, PATCH_START_TOKEN, PATCH_END_TOKEN))
    IMG_TAG_ID, VID_TAG_ID = tokenizer(IMG_TAG_TOKEN, add_special_tokens=False).input_ids[0], tokenizer(VID_TAG_TOKEN, add_special_tokens=False).input_ids[0]
    nl_tokens = tokenizer("\n", add_special_tokens=False).input_ids

    image_indices, images = [], []

    def context_indices(start: int):                                          # :575-585
        b = torch.zeros(1, image_token_length, dtype=torch.int64)
        s = torch.arange(start, start + image_token_length).unsqueeze(0)
        return torch.stack([b, s], dim=0)                                     # [2, 1, image_token_length]

    # ---- image tags (:551-614) ------------------------------------------------------------------
    for batch_idx, input_ids in enumerate(tokens):
        img_positions = [i for i, x in enumerate(input_ids) if x == IMG_TAG_ID]
        if len(img_positions) == 0:
            continue
        if image_path_list is not None:
            assert len(img_positions) == len(image_path_list)
        if image_list is not None:
            assert len(img_positions) == len(image_list)
        new_input_ids, st = [], 0
        for img_idx, img_pos in enumerate(img_positions):
            if image_path_list is not None:
                image_patches, (best_width, best_height) = image_processor.process_images_with_subpatch(image_path_list[img_idx])
            if image_list is not None:
                image_patches, (best_width, best_height) = image_processor.process_images_with_subpatch(image_list[img_idx])
            images.append(image_patches)
            new_input_ids += input_ids[st:img_pos]
            new_input_ids += [IMG_START_ID]
            image_indices.append(context_indices(len(new_input_ids)))
            new_input_ids += [IMG_CONTEXT_ID] * image_token_length
            new_input_ids += [IMG_END_ID]
            if len(image_patches) > 1:
                for _i in range(0, best_height, image_processor.patch_size):
                    new_input_ids += nl_tokens
                    for _j in range(0, best_width, image_processor.patch_size):
                        new_input_ids += [PATCH_START_ID]
                        image_indices.append(context_indices(len(new_input_ids)))
                        new_input_ids += [PATCH_CONTEXT_ID] * image_token_length
                        new_input_ids += [PATCH_END_ID]
            st = img_pos + 1
        new_input_ids += input_ids[st:]
        tokens[batch_idx] = new_input_ids

    # ---- video tags (:619-678) ------------------------------------------------------------------
    for batch_idx, input_ids in enumerate(tokens):
        vid_positions = [i for i, x in enumerate(input_ids) if x == VID_TAG_ID]
        if len(vid_positions) == 0:
            continue
        for lst in (video_path_list, image_path_list, image_list, video_frames_list):
            if lst is not None:
                assert len(vid_positions) == len(lst)
        new_input_ids, st = [], 0
        for vid_idx, vid_pos in enumerate(vid_positions):
            if per_rank:
                video_frames = _video_sources(image_processor, vid_idx, image_list, image_path_list, video_path_list,
                                              video_frames_list, max_num_frame, max_fps)
                pending[len(images)] = video_frames
            elif video_path_list is not None:
                video_frames, _ = image_processor.process_video(video_path_list[vid_idx], max_num_frame, max_fps)
            if not per_rank and image_path_list is not None:
                video_frames = image_processor.process_images([image_path_list[vid_idx]])
            if not per_rank and image_list is not None:
                video_frames = image_processor.process_images([image_list[vid_idx]])
            if not per_rank and video_frames_list is not None:
                video_frames = image_processor.process_images(list(video_frames_list[vid_idx])[:max_num_frame])
            images.append(video_frames)
            new_input_ids += input_ids[st:vid_pos]
            for _ in video_frames:
                new_input_ids += [VID_START_ID]
                image_indices.append(context_indices(len(new_input_ids)))
                new_input_ids += [VID_CONTEXT_ID] * image_token_length
                new_input_ids += [VID_END_ID]
            st = vid_pos + 1
        new_input_ids += input_ids[st:]
        tokens[batch_idx] = new_input_ids

    image_indices = torch.cat(image_indices, dim=1)
    pad_id = tokenizer.pad_token_id if tokenizer.pad_token_id else tokenizer.eos_token_id
    token_lengths = [len(x) for x in tokens]
    tokens = [x + [pad_id] * (-(-len(x) // 64) * 64 - len(x)) for x in tokens]
    if per_rank:
        seq_length = len(tokens[0]) if cp_seq_length is None else (
            cp_seq_length(token_lengths[0]) if callable(cp_seq_length) else cp_seq_length)
        keep = frames_on_this_cp_rank(image_indices[1, :, 0].tolist(), image_token_length, seq_length, cp_size, cp_rank)
        kept, at = [], 0
        for slot, entry in enumerate(images):
            n = len(entry)
            mine = [i for i in range(n) if keep[at + i]]
            at += n
            if not mine:
                continue
            if slot in pending:                                                # only these frames are decoded / uploaded
                kept.append(image_processor.process_images(_open_frames([entry[i] for i in mine])))
            else:
                kept.append(entry[mine])
        assert at == len(keep)
        image_indices = image_indices[:, torch.tensor(keep, dtype=torch.bool)]
        images = kept
    if len(images) == 0:                                                       # a rank whose chunks hold text only
        size = getattr(image_processor, "image_size", 448)
        images = [torch.empty(0, 3, size, size)]
    images = torch.cat(images, dim=0)

    external_inputs = {"indices": image_indices.contiguous().to(device),
                       "images": images.to(dtype=torch.bfloat16 if bf16 else torch.float16).contiguous().to(device)}
    tokens = torch.tensor(tokens, dtype=torch.long, device=device)
    token_lengths = torch.tensor(token_lengths, dtype=torch.long, device=device)
    return external_inputs, tokens, token_lengths


def request_tensors(prompt_ids: Sequence[int], tokens_to_generate: int, tokenizer, image_processor, *, max_generate_length: int = 128,
                    image_list=None, image_path_list=None, video_path_list=None, video_frames_list=None, **kw):
    """One tokenised prompt -> (tokens [1, S], lengths [1], external_inputs or None), the state MegatronModuleForCausalLM.generate
    (module.py:270-360) hands to the decode loop: the prompt is first padded with pad_token_id to prompt + tokens_to_generate
    (or max_generate_length; _tokenize_prompts_and_batch, M/inference/text_generation/tokenization.py:150-166), the media tags
    are expanded on that padded row, and the true context length is the expanded length minus that padding (:357).  Prompt
    templates / text tokenisation stay with the caller.  **kw goes to get_external_inputs (cp_size / cp_rank: per-rank
    loading instead of the reference's rank-0 build + world broadcast, :340-360)."""
    prompt = [int(t) for t in prompt_ids]
    total = len(prompt) + tokens_to_generate if tokens_to_generate > 0 else max_generate_length
    pad_length = total - len(prompt)
    if pad_length < 0:
        raise ValueError("the prompt is longer than max_generate_length")
    device = kw.get("device", "cuda")
    tokens = torch.tensor([prompt + [tokenizer.pad_token_id] * pad_length], dtype=torch.long)
    if image_list is None and image_path_list is None and video_path_list is None and video_frames_list is None:
        return tokens.to(device), torch.tensor([len(prompt)], dtype=torch.long, device=device), None
    external_inputs, tokens, lengths = get_external_inputs(tokens, image_list, image_path_list, video_path_list, tokenizer,
                                                           image_processor, video_frames_list=video_frames_list, **kw)
    return tokens, lengths - pad_length, external_inputs


def generate(model, prompt_ids: Sequence[int], tokens_to_generate: int, tokenizer, image_processor=None, *, do_sample=False,
             top_k=0, top_p=0.0, temperature=1.0, return_output_log_probs=False, use_kv_cache=True, logit_mask=True,
             termination_id=None, reference_compat=False, per_rank_loading=True, **media):
    """The request -> token stream chain of MegatronModuleForCausalLM.generate (module.py:270-402) for one prompt: request_tensors
    on every rank (with per_rank_loading each CP rank builds only its own frames; nothing is broadcast), then the decode loop.
    Yields what generate_tokens_probs_and_return_on_first_stage yields."""
    from . import generation, parallel_state as mpu
    cp = mpu.get_context_parallel_world_size()
    kw = dict(media)
    if cp > 1 and per_rank_loading:
        pad_length = tokens_to_generate if tokens_to_generate > 0 else max(kw.get("max_generate_length", 128) - len(prompt_ids), 0)
        kw.update(cp_size=cp, cp_rank=mpu.get_context_parallel_rank())
        if use_kv_cache:                  # the cached path prefills the prompt alone, padded to the kernels' chunk granularity
            kw["cp_seq_length"] = lambda expanded: generation._cp_prefill_length(expanded - pad_length, cp)
    tokens, lengths, ext = request_tensors(prompt_ids, tokens_to_generate, tokenizer, image_processor, **kw)
    if termination_id is None:
        termination_id = getattr(tokenizer, "eos_token_id", None)
    yield from generation.generate_tokens_probs_and_return_on_first_stage(
        model, tokens, lengths, return_output_log_probs=return_output_log_probs, do_sample=do_sample, top_k=top_k, top_p=top_p,
        temperature=temperature, external_inputs=ext, use_kv_cache=use_kv_cache, logit_mask=logit_mask,
        termination_id=termination_id, reference_compat=reference_compat)
