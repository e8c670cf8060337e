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
    """M/tasks/inference/module.py:693"""
    return torch.tensor(images, dtype=torch.bfloat16)
