"""ImageProcessor — mirror of H/data/processor/image_processor.py (process_images :180-223, the per-frame
hot loop of process_video :136-178) with the arithmetic on the GPU.

The reference pads each frame to a square, resizes it with Pillow (BICUBIC) and normalises it on the rank-0 CPU,
one frame at a time (4096 frames for a 1M-token video), then broadcasts the float tensor.  Here the decoded
uint8 frames go to HBM as they are (3 bytes per pixel) and `vita_frames_resize_norm` produces the
[N, 3, 448, 448] bf16 tensor the ViT consumes.  The only host arithmetic left is Pillow's coefficient table
(`pil_resample_table`, O(448 * taps) doubles per distinct frame size).
"""
from __future__ import annotations

import ctypes as C
import math
from typing import Sequence

import numpy as np
import torch

from . import lib as _L

# long_vita/constants.py:87-92
IMAGENET_DEFAULT_MEAN = [0.485, 0.456, 0.406]
IMAGENET_DEFAULT_STD = [0.229, 0.224, 0.225]
IMAGENET_STANDARD_MEAN = [0.5, 0.5, 0.5]
IMAGENET_STANDARD_STD = [0.5, 0.5, 0.5]
OPENAI_CLIP_MEAN = [0.48145466, 0.4578275, 0.40821073]
OPENAI_CLIP_STD = [0.26862954, 0.26130258, 0.27577711]

PRECISION_BITS = 32 - 8 - 2


def _bicubic(x: float) -> float:
    """Pillow's bicubic_filter (Resample.c), a = -0.5, support 2."""
    a = -0.5
    if x < 0.0:
        x = -x
    if x < 1.0:
        return ((a + 2.0) * x - (a + 3.0)) * x * x + 1
    if x < 2.0:
        return (((x - 5) * x + 8) * x - 4) * a
    return 0.0


def pil_resample_table(in_size: int, out_size: int):
    """Pillow's precompute_coeffs + normalize_coeffs_8bpc for a full-image BICUBIC resize in_size -> out_size:
    (bounds [out, 2] int32, coeffs [out, ksize] int32, ksize).  Same double-precision operations in the same order,
    so the 22-bit fixed-point taps are identical to Pillow's."""
    scale = filterscale = float(in_size) / out_size
    if filterscale < 1.0:
        filterscale = 1.0
    support = 2.0 * filterscale
    ksize = int(math.ceil(support)) * 2 + 1
    bounds = np.zeros((out_size, 2), dtype=np.int32)
    coeffs = np.zeros((out_size, ksize), dtype=np.int32)
    ss = 1.0 / filterscale
    for xx in range(out_size):
        center = 0 + (xx + 0.5) * scale
        xmin = int(center - support + 0.5)
        if xmin < 0:
            xmin = 0
        xmax = int(center + support + 0.5)
        if xmax > in_size:
            xmax = in_size
        xmax -= xmin
        ws = [_bicubic((x + xmin - center + 0.5) * ss) for x in range(xmax)]
        ww = 0.0
        for w in ws:
            ww += w
        for x in range(xmax):
            k = ws[x] / ww if ww != 0.0 else ws[x]
            coeffs[xx, x] = int(-0.5 + k * (1 << PRECISION_BITS)) if k < 0 else int(0.5 + k * (1 << PRECISION_BITS))
        bounds[xx, 0], bounds[xx, 1] = xmin, xmax
    return bounds, coeffs, ksize


class ImageProcessor:
    def __init__(self, process_type="", image_size=448, normalize_type="imagenet", min_patch_grid=1, max_patch_grid=6,
                 device="cuda"):
        self.process_type = process_type
        self.image_size = image_size
        if normalize_type == "imagenet":
            mean, std = IMAGENET_DEFAULT_MEAN, IMAGENET_DEFAULT_STD
        elif normalize_type == "clip":
            mean, std = OPENAI_CLIP_MEAN, OPENAI_CLIP_STD
        elif normalize_type == "siglip":
            mean, std = IMAGENET_STANDARD_MEAN, IMAGENET_STANDARD_STD
        else:
            raise NotImplementedError
        self.mean, self.std = mean, std
        self.patch_size = image_size
        self.min_patch_grid, self.max_patch_grid = min_patch_grid, max_patch_grid
        self.device = device
        self._tables = {}
        if process_type == "anyres":                                    # :48-59
            self.grid_pinpoints = [(i, j) for i in range(min_patch_grid, max_patch_grid + 1)
                                   for j in range(min_patch_grid, max_patch_grid + 1)]
            self.possible_resolutions = [[dim * self.patch_size for dim in pair] for pair in self.grid_pinpoints]
        if process_type == "dynamic":                                   # :61-77
            max_num, min_num = self.max_patch_grid, self.min_patch_grid
            target_ratios = set((i, j) for n in range(min_num, max_num + 1) for i in range(1, n + 1)
                                for j in range(1, n + 1) if i * j <= max_num and i * j >= min_num)
            self.target_ratios = sorted(target_ratios, key=lambda x: x[0] * x[1])
            self.possible_resolutions = [[dim * self.patch_size for dim in pair] for pair in self.target_ratios]

    def _table(self, in_size: int, out_size: int):
        t = self._tables.get((in_size, out_size))
        if t is None:
            b, c, k = pil_resample_table(in_size, out_size)
            t = (torch.from_numpy(b).to(self.device), torch.from_numpy(c).to(self.device), k)
            self._tables[(in_size, out_size)] = t
        return t

    def _resize_norm(self, frames: torch.Tensor, out_w: int, out_h: int, tile: int, pad_to_square: bool, return_u8=False,
                     canvas=None):
        """frames [N, H, W, 3] uint8 (device) -> [N * blocks, 3, tile, tile] bf16: Pillow-exact resize to out_w x out_h,
        cut into tile x tile blocks (row-major), normalised.  canvas = (canvas_w, canvas_h, off_x, off_y, rgb): the resized
        image is pasted onto a canvas of that size filled with `rgb` first (resize_and_pad_image)."""
        if frames.dtype != torch.uint8 or frames.dim() != 4 or frames.shape[-1] != 3 or not frames.is_cuda:
            raise ValueError("frames must be a [N, H, W, 3] uint8 HIP device tensor (no CPU fallback)")
        frames = frames.contiguous()
        n, h, w, _ = frames.shape
        P = max(h, w)
        pw, ph = (P, P) if pad_to_square else (w, h)
        hb, hc, hk = self._table(pw, out_w)
        vb, vc, vk = self._table(ph, out_h)
        cw, ch, ox, oy = (canvas[0], canvas[1], canvas[2], canvas[3]) if canvas else (0, 0, 0, 0)
        blocks = ((cw if canvas else out_w) // tile) * ((ch if canvas else out_h) // tile)
        out = torch.empty(n * blocks, 3, tile, tile, dtype=torch.bfloat16, device=frames.device)
        if canvas:                                                       # normalised pad colour, same float32 op order
            for ci in range(3):
                v = (np.float32(canvas[4][ci]) * np.float32(1.0) / np.float32(255.0) - np.float32(self.mean[ci])) / np.float32(self.std[ci])
                out[:, ci].fill_(float(v))
        u8 = torch.empty(n, out_h, out_w, 3, dtype=torch.uint8, device=frames.device) if return_u8 else None
        pad = (C.c_int * 3)(*[int(x * 255) for x in self.mean])             # tuple(int(x * 255) for x in mean) :204
        mean = (C.c_float * 3)(*[float(np.float32(x)) for x in self.mean])
        std = (C.c_float * 3)(*[float(np.float32(x)) for x in self.std])
        chunk = max(1, min(n, 65535, (1 << 30) // (ph * out_w * 3)))       # bound the uint8 scratch to 1 GiB
        tmp = torch.empty(chunk * ph * out_w * 3, dtype=torch.uint8, device=frames.device)
        st = torch.cuda.current_stream().cuda_stream
        for i in range(0, n, chunk):
            m = min(chunk, n - i)
            _L.check(_L.load().vita_frames_resize_norm(
                frames[i].data_ptr(), h * w * 3, m, h, w, int(pad_to_square), pad, out_w, out_h, tile, cw, ch, ox, oy, hb.data_ptr(),
                hc.data_ptr(), hk, vb.data_ptr(), vc.data_ptr(), vk, mean, std, tmp.data_ptr(), out[i * blocks].data_ptr(),
                None if u8 is None else u8[i].data_ptr(), st), "vita_frames_resize_norm")
        return (out, u8) if return_u8 else out

    def process_frames(self, frames: torch.Tensor, return_u8: bool = False):
        """frames [N, H, W, 3] uint8 on the device -> [N, 3, S, S] bf16 (and optionally the uint8 resize): the per-frame
        body of process_images (:203-221): expand2square, BICUBIC resize to S x S, normalise."""
        S = self.image_size
        return self._resize_norm(frames, S, S, S, True, return_u8)

    # -- dynamic_preprocess (:404-448) + process_dynamic (:299-316) ---------------------------------------------------
    def find_closest_aspect_ratio(self, aspect_ratio, width, height):
        """:386-401"""
        best_ratio_diff, best_ratio, area = float("inf"), (1, 1), width * height
        for ratio in self.target_ratios:
            target_aspect_ratio = ratio[0] / ratio[1]
            ratio_diff = abs(aspect_ratio - target_aspect_ratio)
            if ratio_diff < best_ratio_diff:
                best_ratio_diff, best_ratio = ratio_diff, ratio
            elif ratio_diff == best_ratio_diff:
                if area > 0.5 * self.image_size * self.image_size * ratio[0] * ratio[1]:
                    best_ratio = ratio
        return best_ratio

    def process_dynamic(self, img_or_array):
        """One image -> ([thumbnail +] tiles [B, 3, S, S] bf16 on the device, (target_width, target_height)): the image is
        resized (BICUBIC) to the closest-aspect grid of S x S tiles and cut up; with more than one tile a S x S thumbnail of
        the whole image goes first (use_thumbnail=True, :312).  The tiles are square already, so process_images' own
        expand2square / resize are identities and only the normalisation remains."""
        if self.process_type != "dynamic":
            raise ValueError("process_dynamic needs process_type='dynamic'")
        arr = np.asarray(img_or_array.convert("RGB") if hasattr(img_or_array, "convert") else img_or_array, dtype=np.uint8)
        h, w, _ = arr.shape
        S = self.image_size
        ratio = self.find_closest_aspect_ratio(w / h, w, h)
        tw, th = S * ratio[0], S * ratio[1]
        frame = torch.from_numpy(arr)[None].to(self.device)
        tiles = self._resize_norm(frame, tw, th, S, False)
        if tiles.shape[0] != 1:
            tiles = torch.cat([self._resize_norm(frame, S, S, S, False), tiles], dim=0)
        return tiles, (tw, th)

    # -- select_best_resolution (:319-352) + resize_and_pad_image (:355-394) + divide_to_patches (:397-416) + process_anyres ----
    def select_best_resolution(self, original_size):
        original_width, original_height = original_size
        best_fit, max_effective_resolution, min_wasted_resolution = None, 0, float("inf")
        for width, height in self.possible_resolutions:
            scale = min(width / original_width, height / original_height)
            downscaled_width, downscaled_height = int(original_width * scale), int(original_height * scale)
            effective_resolution = min(downscaled_width * downscaled_height, original_width * original_height)
            wasted_resolution = (width * height) - effective_resolution
            if effective_resolution > max_effective_resolution or (
                    effective_resolution == max_effective_resolution and wasted_resolution < min_wasted_resolution):
                max_effective_resolution, min_wasted_resolution, best_fit = effective_resolution, wasted_resolution, (width, height)
        return best_fit

    def process_anyres(self, img_or_array):
        """One image -> ([image, tiles...] [1 + B, 3, S, S] bf16, best_resolution): aspect-preserving BICUBIC resize into the
        best grid resolution, black padding, S x S tiles; the whole image (expand2square + resize) goes first.  A 1 x 1 grid
        returns just the image (:279-282)."""
        if self.process_type != "anyres":
            raise ValueError("process_anyres needs process_type='anyres'")
        arr = np.asarray(img_or_array.convert("RGB") if hasattr(img_or_array, "convert") else img_or_array, dtype=np.uint8)
        h, w, _ = arr.shape
        S = self.image_size
        target_width, target_height = self.select_best_resolution((w, h))
        frame = torch.from_numpy(arr)[None].to(self.device)
        whole = self._resize_norm(frame, S, S, S, True)
        if (target_width, target_height) == (S, S):
            return whole, (target_width, target_height)
        scale_w, scale_h = target_width / w, target_height / h              # resize_and_pad_image :338-350
        if scale_w < scale_h:
            new_width, new_height = target_width, min(math.ceil(h * scale_w), target_height)
        else:
            new_height, new_width = target_height, min(math.ceil(w * scale_h), target_width)
        paste_x, paste_y = (target_width - new_width) // 2, (target_height - new_height) // 2
        tiles = self._resize_norm(frame, new_width, new_height, S, False,
                                  canvas=(target_width, target_height, paste_x, paste_y, (0, 0, 0)))
        return torch.cat([whole, tiles], dim=0), (target_width, target_height)

    def process_images_with_subpatch(self, img_or_path):
        """:225-240"""
        if self.process_type == "anyres":
            return self.process_anyres(img_or_path)
        if self.process_type == "dynamic":
            return self.process_dynamic(img_or_path)
        return self.process_images([img_or_path])

    # -- process_video (:136-178) ----------------------------------------------------------------------------------------
    @staticmethod
    def video_frame_indices(total_frames: int, avg_fps: float, max_fps: int = 1, num_frames: int = 8):
        """Frame selection of get_video_frames (:113-124) for a decoded-on-demand video: step = max(fps / max_fps,
        total / (num_frames + 1)), indices int(i * step) below total."""
        step_size = total_frames / (num_frames + 1)
        step_size = max(avg_fps / max_fps, step_size)
        indices = [int(i * step_size) for i in range(0, num_frames)]
        return [i for i in indices if i < total_frames]

    @staticmethod
    def directory_frame_paths(video_dir: str, max_num_frame: int = 8, max_fps: int = 1):
        """Frame selection of the directory branch (:137-163): png / jpeg / jpg files, natural order (natsort.natsorted:
        digit runs compare as integers), fps 2 for ShareGPTVideo else 1, target = int(min(total / fps * max_fps, max)),
        every int(total / target)-th file."""
        import os
        import re
        all_filepath = []
        for root, _dirs, files in os.walk(video_dir):
            for filename in files:
                if filename.endswith("png") or filename.endswith("jpeg") or filename.endswith("jpg"):
                    all_filepath.append(os.path.join(root, filename))
        if len(all_filepath) == 0:
            return None
        all_filepath.sort(key=lambda sp: [int(t) if t.isdigit() else t for t in re.split(r"(\d+)", sp)])
        total_frame = len(all_filepath)
        fps = 2 if "ShareGPTVideo" in video_dir else 1
        target_frame = int(min(total_frame / fps * max_fps, max_num_frame))
        index = [int(1.0 * total_frame / target_frame) * x for x in range(target_frame)]
        return [all_filepath[x] for x in index]

    def process_video(self, video_file_or_dir, max_num_frame=8, max_fps=1):
        """Directory of frames: select (above), decode with PIL on the host, preprocess on the GPU.  A video FILE needs a
        decoder (decord in the reference): decode the frames chosen by video_frame_indices yourself and call
        process_images / process_frames."""
        import os

        from PIL import Image
        if os.path.isdir(video_file_or_dir):
            paths = self.directory_frame_paths(video_file_or_dir, max_num_frame, max_fps)
            if paths is None:
                return None
            return self.process_images([Image.open(x).convert("RGB") for x in paths]), paths
        raise NotImplementedError("video files need a decoder (decord is not part of this framework): see video_frame_indices")

    def process_images(self, img_or_array_list: Sequence):
        """:180-223.  Accepts decoded frames (PIL images or [H, W, 3] uint8 arrays); frames of equal size are batched
        into one launch.  Returns [N, 3, S, S] bf16 on the device (the dtype module.py:693 casts to)."""
        arrays = [np.asarray(x.convert("RGB") if hasattr(x, "convert") else x, dtype=np.uint8) for x in img_or_array_list]
        out = torch.empty(len(arrays), 3, self.image_size, self.image_size, dtype=torch.bfloat16, device=self.device)
        groups = {}
        for i, a in enumerate(arrays):
            groups.setdefault(a.shape, []).append(i)
        for shape, idx in groups.items():
            batch = torch.from_numpy(np.stack([arrays[i] for i in idx])).to(self.device)
            out[torch.tensor(idx, device=self.device)] = self.process_frames(batch)
        return out
