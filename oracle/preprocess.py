"""TEST INFRASTRUCTURE ONLY (see oracle/__init__.py) — CPU restatement of the frame preprocessing step,
ImageProcessor.process_images, H/data/processor/image_processor.py:180-223, plus the bf16 cast of
M/tasks/inference/module.py:693.  Pillow (a dependency of the reference, present in this image) does the
resize exactly as the reference calls it; pinned by tests/golden/image_processor.pt, which was produced by
the reference's own class (oracle/make_golden.py:golden_image_processor)."""
import numpy as np
import torch
from PIL import Image

MEANS = {"imagenet": ([0.485, 0.456, 0.406], [0.229, 0.224, 0.225]),          # long_vita/constants.py:87-92
         "siglip": ([0.5, 0.5, 0.5], [0.5, 0.5, 0.5]),
         "clip": ([0.48145466, 0.4578275, 0.40821073], [0.26862954, 0.26130258, 0.27577711])}


def expand2square(pil_img, background_color):
    """:189-201"""
    width, height = pil_img.size
    if width == height:
        return pil_img
    if width > height:
        result = Image.new(pil_img.mode, (width, width), background_color)
        result.paste(pil_img, (0, (width - height) // 2))
        return result
    result = Image.new(pil_img.mode, (height, height), background_color)
    result.paste(pil_img, ((height - width) // 2, 0))
    return result


def resize_u8(frame: np.ndarray, image_size: int, mean) -> np.ndarray:
    """[H, W, 3] uint8 -> [S, S, 3] uint8: pad to square with int(mean * 255) (:204) and BICUBIC resize (:206-208)."""
    img = expand2square(Image.fromarray(frame), tuple(int(x * 255) for x in mean))
    img = img.resize((image_size, image_size), resample=Image.Resampling.BICUBIC)
    return np.array(img)


def process_images(frames, image_size=448, normalize_type="imagenet") -> torch.Tensor:
    """list of [H, W, 3] uint8 arrays -> [N, 3, S, S] float32 (:203-221)."""
    mean, std = MEANS[normalize_type]
    out = torch.ones([len(frames), 3, image_size, image_size])
    for i, f in enumerate(frames):
        image = np.array(resize_u8(np.asarray(f), image_size, mean), dtype=np.float32)
        image = image * 1.0 / 255.0                                              # :211
        image = (image - np.array(mean, dtype=image.dtype)) / np.array(std, dtype=image.dtype)   # :213-215
        out[i] = torch.tensor(image, dtype=torch.float32).permute(2, 0, 1)       # :217-218
    return out


def to_model_dtype(images: torch.Tensor) -> torch.Tensor:
    """M/tasks/inference/module.py:693 (`torch.tensor(images, dtype=torch.bfloat16)`: a copy with the RNE cast)"""
    return images.detach().clone().to(torch.bfloat16) if isinstance(images, torch.Tensor) else torch.tensor(images, dtype=torch.bfloat16)


def dynamic_preprocess(frame: np.ndarray, min_num=1, max_num=12, image_size=448, use_thumbnail=True):
    """H/data/processor/image_processor.py:386-448 (find_closest_aspect_ratio + dynamic_preprocess): closest-aspect grid of
    image_size tiles, BICUBIC resize (PIL's default), crop in row-major order, thumbnail first if more than one tile."""
    image = Image.fromarray(frame)
    orig_width, orig_height = image.size
    aspect_ratio = orig_width / orig_height
    target_ratios = set((i, j) for n in range(min_num, max_num + 1) for i in range(1, n + 1) for j in range(1, n + 1)
                        if i * j <= max_num and i * j >= min_num)
    target_ratios = sorted(target_ratios, key=lambda x: x[0] * x[1])
    best_ratio_diff, best_ratio, area = float("inf"), (1, 1), orig_width * orig_height
    for ratio in target_ratios:
        ratio_diff = abs(aspect_ratio - ratio[0] / ratio[1])
        if ratio_diff < best_ratio_diff:
            best_ratio_diff, best_ratio = ratio_diff, ratio
        elif ratio_diff == best_ratio_diff and area > 0.5 * image_size * image_size * ratio[0] * ratio[1]:
            best_ratio = ratio
    target_width, target_height = image_size * best_ratio[0], image_size * best_ratio[1]
    blocks = best_ratio[0] * best_ratio[1]
    resized_img = image.resize((target_width, target_height))
    per_row = target_width // image_size
    tiles = [resized_img.crop(((i % per_row) * image_size, (i // per_row) * image_size,
                               (i % per_row + 1) * image_size, (i // per_row + 1) * image_size)) for i in range(blocks)]
    if use_thumbnail and len(tiles) != 1:
        tiles = [image.resize((image_size, image_size))] + tiles
    return [np.array(t) for t in tiles], (target_width, target_height)


def process_dynamic(frame: np.ndarray, image_size=448, normalize_type="imagenet", min_patch_grid=1, max_patch_grid=12):
    """ImageProcessor.process_dynamic (:299-316): tiles -> process_images -> [B, 3, S, S] float32, (tw, th)."""
    tiles, res = dynamic_preprocess(frame, min_patch_grid, max_patch_grid, image_size, True)
    return process_images(tiles, image_size, normalize_type), res


def process_anyres(frame: np.ndarray, image_size=448, normalize_type="imagenet", min_patch_grid=1, max_patch_grid=6):
    """ImageProcessor.process_anyres (:242-266) with select_best_resolution (:319-352), resize_and_pad_image (:355-394) and
    divide_to_patches (:397-416): [image + tiles, 3, S, S] float32, best_resolution."""
    import math
    image = Image.fromarray(frame)
    possible = [[i * image_size, j * image_size] for i in range(min_patch_grid, max_patch_grid + 1)
                for j in range(min_patch_grid, max_patch_grid + 1)]
    ow, oh = image.size
    best_fit, max_eff, min_waste = None, 0, float("inf")
    for width, height in possible:
        scale = min(width / ow, height / oh)
        dw, dh = int(ow * scale), int(oh * scale)
        eff = min(dw * dh, ow * oh)
        waste = width * height - eff
        if eff > max_eff or (eff == max_eff and waste < min_waste):
            max_eff, min_waste, best_fit = eff, waste, (width, height)
    tw, th = best_fit
    scale_w, scale_h = tw / ow, th / oh
    if scale_w < scale_h:
        nw, nh = tw, min(math.ceil(oh * scale_w), th)
    else:
        nh, nw = th, min(math.ceil(ow * scale_h), tw)
    new_image = Image.new("RGB", (tw, th), (0, 0, 0))
    new_image.paste(image.resize((nw, nh)), ((tw - nw) // 2, (th - nh) // 2))
    patches = [np.array(new_image.crop((j, i, j + image_size, i + image_size)))
               for i in range(0, th, image_size) for j in range(0, tw, image_size)]
    tiles = [frame] if best_fit == (image_size, image_size) else [frame] + patches
    return process_images(tiles, image_size, normalize_type), best_fit
