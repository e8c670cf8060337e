"""Secondary measurement: frames/s of vita_frames_resize_norm (720p RGB -> 448x448 bf16)
beside the reference's per-frame Pillow + numpy loop on the host.  One JSON line."""
import argparse, json, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import torch
from long_vita_amd import image_processor, lib
from oracle import preprocess as opre           # cpu_baseline leg only

ap = argparse.ArgumentParser()
ap.add_argument("--frames", type=int, default=512)
ap.add_argument("--height", type=int, default=720)
ap.add_argument("--width", type=int, default=1280)
ap.add_argument("--cpu-frames", type=int, default=16)
a = ap.parse_args()
lib.load(allow_build=False)
rng = np.random.default_rng(0)
frames = torch.from_numpy(rng.integers(0, 256, (a.frames, a.height, a.width, 3), dtype=np.uint8)).cuda()
proc = image_processor.ImageProcessor("", 448, "imagenet")
out = proc.process_frames(frames)
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(5):
    out = proc.process_frames(frames)
e1.record()
torch.cuda.synchronize()
ms = e0.elapsed_time(e1) / 5
P = max(a.height, a.width)
alg = a.frames * (a.height * a.width * 3 + 2 * P * 448 * 3 + 3 * 448 * 448 * 2)
t0 = time.perf_counter()
opre.to_model_dtype(opre.process_images(list(frames[: a.cpu_frames].cpu().numpy()), 448, "imagenet"))
cpu_s = (time.perf_counter() - t0) / a.cpu_frames
print(json.dumps({"what": "frame preprocessing: expand2square + Pillow-exact bicubic + normalise -> bf16",
                  "frames": a.frames, "height": a.height, "width": a.width, "ms": ms, "frames_per_s": a.frames / ms * 1e3,
                  "algorithmic_GBps": alg / ms / 1e6, "cpu_reference_ms_per_frame": cpu_s * 1e3,
                  "cpu_frames_per_s_1core": 1 / cpu_s}))
