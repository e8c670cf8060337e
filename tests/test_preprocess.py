"""Frame preprocessing: oracle vs the fixture made by the reference's own ImageProcessor, the
host-side Pillow coefficient table, and (gpu) the HIP kernels — all bit-exact (integer resize, float32 normalise,
round-to-nearest-even bf16)."""
import numpy as np
import pytest
import torch

from conftest import load_golden
from oracle import preprocess as opre

SHAPES = [(720, 1280, 448), (480, 640, 448), (100, 37, 56), (448, 448, 448), (300, 300, 448), (33, 500, 56),
          (449, 447, 448), (1080, 1920, 448)]


def test_oracle_matches_reference_fixture():
    g = load_golden("image_processor.pt")
    for case in g["cases"]:
        frames = [f.numpy() for f in case["frames"]]
        out = opre.process_images(frames, case["image_size"], case["normalize_type"])
        assert torch.equal(out, case["output"])
        assert torch.equal(opre.to_model_dtype(out), case["output_bf16"])


def test_oracle_anyres_tiling_matches_reference_fixture():
    g = load_golden("image_processor.pt")
    assert len(g["anyres"]) == 5
    for case in g["anyres"]:
        out, res = opre.process_anyres(case["frame"].numpy(), 28, "imagenet", 1, 4)
        assert tuple(res) == case["resolution"] and torch.equal(opre.to_model_dtype(out), case["output_bf16"])


def test_oracle_dynamic_tiling_matches_reference_fixture():
    g = load_golden("image_processor.pt")
    assert len(g["dynamic"]) == 5
    for case in g["dynamic"]:
        out, res = opre.process_dynamic(case["frame"].numpy(), 28, "imagenet", 1, 12)
        assert res == case["resolution"] and torch.equal(opre.to_model_dtype(out), case["output_bf16"])


@pytest.mark.parametrize("H,W,S", SHAPES[:7])
def test_host_coefficient_table_reproduces_pillow(H, W, S):
    """pil_resample_table + the two integer passes (numpy restatement of the kernels) == Pillow's resize."""
    from long_vita_amd.image_processor import IMAGENET_DEFAULT_MEAN, pil_resample_table
    rng = np.random.default_rng(H * 7 + W)
    img = rng.integers(0, 256, (H, W, 3), dtype=np.uint8)
    P = max(H, W)
    sq = np.empty((P, P, 3), np.uint8)
    sq[:] = np.array([int(x * 255) for x in IMAGENET_DEFAULT_MEAN], dtype=np.uint8)
    oy, ox = (P - H) // 2, (P - W) // 2
    sq[oy: oy + H, ox: ox + W] = img
    b, c, k = pil_resample_table(P, S)
    assert c.shape == (S, k) and int(b[:, 1].max()) <= k
    tmp = np.empty((P, S, 3), np.uint8)
    for xx in range(S):
        x0, n = b[xx]
        acc = (1 << 21) + (sq[:, x0: x0 + n].astype(np.int64) * c[xx, :n, None]).sum(1)
        tmp[:, xx] = np.clip(acc >> 22, 0, 255)
    res = np.empty((S, S, 3), np.uint8)
    for yy in range(S):
        y0, n = b[yy]
        acc = (1 << 21) + (tmp[y0: y0 + n].astype(np.int64) * c[yy, :n, None, None]).sum(0)
        res[yy] = np.clip(acc >> 22, 0, 255)
    assert np.array_equal(res, opre.resize_u8(img, S, IMAGENET_DEFAULT_MEAN))


def test_host_coefficient_table_random_sizes_vs_pillow():
    """60 random (input size, output size) pairs from 1 px to 4K, up- and down-scaling: one horizontal pass with the host
    table (bounds, 22-bit taps) equals Pillow's BICUBIC resize of a random row bit for bit — so the tap table, the only
    host arithmetic of vita_frames_resize_norm, is Pillow's for every geometry, not only the tested frame sizes."""
    from PIL import Image

    from long_vita_amd.image_processor import pil_resample_table
    rng = np.random.default_rng(2024)
    sizes = [(1, 1), (1, 7), (7, 1), (2, 448), (448, 448), (4096, 448), (3, 2), (449, 448), (447, 448)]
    while len(sizes) < 60:
        a = int(rng.choice([rng.integers(1, 64), rng.integers(64, 1200), rng.integers(1200, 4097)]))
        b = int(rng.choice([rng.integers(1, 64), rng.integers(64, 1200)]))
        sizes.append((a, b))
    for n_in, n_out in sizes:
        row = rng.integers(0, 256, (3, n_in, 3), dtype=np.uint8)               # 3 image rows, so only the width is resized
        want = np.asarray(Image.fromarray(row).resize((n_out, 3), Image.BICUBIC))
        b, c, k = pil_resample_table(n_in, n_out)
        got = np.empty((3, n_out, 3), np.uint8)
        for xx in range(n_out):
            x0, n = b[xx]
            acc = (1 << 21) + (row[:, x0: x0 + n].astype(np.int64) * c[xx, :n, None]).sum(1)
            got[:, xx] = np.clip(acc >> 22, 0, 255)
        assert np.array_equal(got, want), (n_in, n_out)


# ---------------------------------------------------------------------------------------------
@pytest.fixture(scope="module")
def proc_mod():
    from long_vita_amd import image_processor, lib
    lib.load(allow_build=False)
    return image_processor


@pytest.mark.gpu
@pytest.mark.parametrize("H,W,S", SHAPES)
@pytest.mark.parametrize("norm", ["imagenet", "clip"])
def test_hip_frames_bit_exact_vs_oracle(proc_mod, H, W, S, norm):
    if norm == "clip" and S == 448 and H > 500:
        pytest.skip("one normalisation is enough for the large frames")
    rng = np.random.default_rng(H + W)
    n = 3
    frames = rng.integers(0, 256, (n, H, W, 3), dtype=np.uint8)
    frames[1] = (np.indices((H, W)).sum(0)[..., None] * np.array([3, 2, 1]) % 256).astype(np.uint8)   # smooth ramps
    proc = proc_mod.ImageProcessor("", image_size=S, normalize_type=norm)
    out, u8 = proc.process_frames(torch.from_numpy(frames).cuda(), return_u8=True)
    mean = opre.MEANS[norm][0]
    ref_u8 = np.stack([opre.resize_u8(f, S, mean) for f in frames])
    assert np.array_equal(u8.cpu().numpy(), ref_u8)                                   # integer resize: bit-exact
    ref = opre.to_model_dtype(opre.process_images(list(frames), S, norm))
    assert out.shape == ref.shape and torch.equal(out.cpu(), ref)                     # fp32 normalise + RNE bf16: bit-exact


@pytest.mark.gpu
def test_hip_process_images_on_reference_fixture_and_mixed_sizes(proc_mod):
    g = load_golden("image_processor.pt")
    for case in g["cases"]:
        proc = proc_mod.ImageProcessor("", image_size=case["image_size"], normalize_type=case["normalize_type"])
        out = proc.process_images([f.numpy() for f in case["frames"]])                # frames of different sizes
        assert torch.equal(out.cpu(), case["output_bf16"])
    with pytest.raises(ValueError):
        proc.process_frames(torch.zeros(1, 8, 8, 3, dtype=torch.uint8))               # CPU tensor: no fallback


@pytest.mark.gpu
def test_hip_dynamic_tiling_bit_exact(proc_mod):
    """process_dynamic on the device: closest-aspect grid, non-square Pillow-exact resize, tile cut, thumbnail first — equal to
    the fixture made by the reference's class (image_size 28) and to the oracle at the real size (448, max_patch_grid 12)."""
    g = load_golden("image_processor.pt")
    proc = proc_mod.ImageProcessor("dynamic", image_size=28, normalize_type="imagenet", max_patch_grid=12)
    for case in g["dynamic"]:
        tiles, res = proc.process_dynamic(case["frame"].numpy())
        assert tuple(res) == case["resolution"] and torch.equal(tiles.cpu(), case["output_bf16"])
    proc = proc_mod.ImageProcessor("dynamic", image_size=448, normalize_type="imagenet", max_patch_grid=12)
    rng = np.random.default_rng(3)
    for h, w in [(720, 1280), (1000, 333), (448, 448), (500, 1400)]:
        frame = rng.integers(0, 256, (h, w, 3), dtype=np.uint8)
        tiles, res = proc.process_images_with_subpatch(frame)
        ref, ref_res = opre.process_dynamic(frame, 448, "imagenet", 1, 12)
        assert tuple(res) == ref_res and torch.equal(tiles.cpu(), opre.to_model_dtype(ref))


def test_video_frame_selection_rules_match_reference_fixture(tmp_path):
    """Which frames of a video are used (get_video_frames :113-124, the directory branch of process_video :137-163): same
    indices / file names as the reference's class picked under a fake decord reader and natural file order."""
    from PIL import Image

    from long_vita_amd.image_processor import ImageProcessor
    g = load_golden("image_processor.pt")
    assert len(g["video_index_rule"]) == 5 and len(g["video_dir_rule"]) == 3
    for c in g["video_index_rule"]:
        idx = ImageProcessor.video_frame_indices(c["total"], c["fps"], c["max_fps"], c["num_frames"])
        assert [i % 256 for i in idx] == c["picked"]
    for c in g["video_dir_rule"]:
        d = tmp_path / c["tag"]
        d.mkdir()
        for i in range(c["nfiles"]):
            Image.fromarray(np.full((4, 4, 3), i, dtype=np.uint8)).save(str(d / f"frame{i}.png"))
        paths = ImageProcessor.directory_frame_paths(str(d), c["max_num_frame"], 1)
        assert [p.split("/")[-1] for p in paths] == c["picked"]


@pytest.mark.gpu
def test_hip_anyres_tiling_bit_exact(proc_mod):
    """process_anyres on the device: best grid resolution, aspect-preserving resize pasted onto a black canvas, tile cut, the
    whole image first — equal to the reference-made fixture (image_size 28) and to the oracle at 448."""
    g = load_golden("image_processor.pt")
    proc = proc_mod.ImageProcessor("anyres", image_size=28, normalize_type="imagenet", max_patch_grid=4)
    for case in g["anyres"]:
        tiles, res = proc.process_anyres(case["frame"].numpy())
        assert tuple(res) == case["resolution"] and torch.equal(tiles.cpu(), case["output_bf16"])
    proc = proc_mod.ImageProcessor("anyres", image_size=448, normalize_type="imagenet", max_patch_grid=3)
    rng = np.random.default_rng(5)
    for h, w in [(720, 1280), (1000, 333), (300, 300)]:
        frame = rng.integers(0, 256, (h, w, 3), dtype=np.uint8)
        tiles, res = proc.process_images_with_subpatch(frame)
        ref, ref_res = opre.process_anyres(frame, 448, "imagenet", 1, 3)
        assert tuple(res) == tuple(ref_res) and torch.equal(tiles.cpu(), opre.to_model_dtype(ref))
