"""Generate tests/golden/*.pt by running the REFERENCE'S OWN PYTHON (read-only /root/reference)
in this container.  Run here only — /root/reference does not exist on the GPU box:

    python -m oracle.make_golden

The Megatron-side modules import `megatron.*` (absent); they are imported under
permissive stub modules so that their first-party function bodies run unmodified on CPU.
`torch.cuda.current_device()` / `.cuda()` / `device='cuda'` are redirected to CPU for the
duration.  Only small tensors are stored (fixtures stay < 1 MB each); big inputs are re-derived
from seeds by the tests.  TEST INFRASTRUCTURE ONLY.
"""
from __future__ import annotations

import contextlib
import importlib
import os
import sys
import types
from unittest import mock

import torch

REF = "/root/reference"
OUT = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests", "golden")


# ------------------------------------------------------------------------------------------------
# stub environment
# ------------------------------------------------------------------------------------------------
class _Args:
    pass


STATE = {"args": _Args(), "cp_size": 1, "cp_rank": 0}


class _StubModule(types.ModuleType):
    """Module whose unknown attributes are MagicMocks (so `from megatron.x import Y` succeeds)."""

    def __getattr__(self, name):
        if name.startswith("__"):
            raise AttributeError(name)
        m = mock.MagicMock(name=f"{self.__name__}.{name}")
        setattr(self, name, m)
        return m


class _StubFinder:
    """Fabricates any missing `megatron.*` / `apex.*` module on demand (an attribute-mocking package), so that reference files
    whose module-level imports reach deep into Megatron-LM can be imported for the one function a fixture needs."""

    ROOTS = ("megatron", "apex", "mindspeed", "flash_attn", "transformer_engine")

    def find_spec(self, fullname, path=None, target=None):
        import importlib.machinery
        if fullname.split(".")[0] not in self.ROOTS:
            return None
        return importlib.machinery.ModuleSpec(fullname, self, is_package=True)

    def create_module(self, spec):
        m = _StubModule(spec.name)
        m.__path__ = []
        return m

    def exec_module(self, module):
        pass


def _install_stubs():
    if not any(isinstance(f, _StubFinder) for f in sys.meta_path):
        sys.meta_path.append(_StubFinder())
    names = [
        "megatron", "megatron.training", "megatron.core", "megatron.core.mpu", "megatron.core.parallel_state",
        "megatron.core.tensor_parallel", "megatron.core.tensor_parallel.mappings",
        "megatron.core.tensor_parallel.utils", "megatron.core.tensor_parallel.random",
        "megatron.core.model_parallel_config", "megatron.core.utils", "megatron.core.transformer",
        "megatron.core.transformer.module", "megatron.core.transformer.transformer_config",
        "megatron.core.transformer.utils", "megatron.core.dist_checkpointing",
        "megatron.core.dist_checkpointing.mapping", "megatron.legacy", "megatron.legacy.model",
        "megatron.legacy.model.module", "apex", "apex.multi_tensor_apply", "amp_C",
    ]
    for n in names:
        if n not in sys.modules:
            sys.modules[n] = _StubModule(n)
            sys.modules[n].__path__ = []                  # a package: deeper submodules come from _StubFinder
    for n in names:
        if "." in n:
            parent, child = n.rsplit(".", 1)
            setattr(sys.modules[parent], child, sys.modules[n])
    sys.modules["megatron.training"].get_args = lambda: STATE["args"]
    for mod in ("megatron.core.mpu", "megatron.core.parallel_state"):
        m = sys.modules[mod]
        m.get_context_parallel_world_size = lambda: STATE["cp_size"]
        m.get_context_parallel_rank = lambda: STATE["cp_rank"]
        m.get_tensor_model_parallel_world_size = lambda: 1
        m.get_tensor_model_parallel_rank = lambda: 0
    sys.modules["megatron.core"].mpu = sys.modules["megatron.core.mpu"]
    sys.modules["megatron.core"].parallel_state = sys.modules["megatron.core.parallel_state"]

    class MegatronModule(torch.nn.Module):
        def __init__(self, config=None):
            super().__init__()
            self.config = config

    sys.modules["megatron.core.transformer.module"].MegatronModule = MegatronModule

    def prepare_input_tensors_for_wgrad_compute(grad_output, all_gathered_input):
        # Megatron-LM core_r0.7.0 megatron/core/utils.py (published behaviour): flatten [s, b, h] -> [s*b, h]
        if grad_output.dim() == 3:
            grad_output = grad_output.reshape(grad_output.shape[0] * grad_output.shape[1], grad_output.shape[2])
            all_gathered_input = all_gathered_input.reshape(
                all_gathered_input.shape[0] * all_gathered_input.shape[1], all_gathered_input.shape[2])
        return grad_output, all_gathered_input

    sys.modules["megatron.core.utils"].prepare_input_tensors_for_wgrad_compute = prepare_input_tensors_for_wgrad_compute
    if REF not in sys.path:
        sys.path.insert(0, REF)


@contextlib.contextmanager
def cpu_as_cuda():
    """Redirect the reference's hard-coded CUDA placement to CPU."""
    real_arange, real_tensor, real_zeros = torch.arange, torch.tensor, torch.zeros

    def strip(fn):
        def wrapped(*a, **k):
            if str(k.get("device", "")).startswith("cuda") or isinstance(k.get("device"), int):
                k["device"] = "cpu"
            k.pop("pin_memory", None)
            return fn(*a, **k)
        return wrapped

    patches = [
        mock.patch.object(torch, "arange", strip(real_arange)),
        mock.patch.object(torch, "tensor", strip(real_tensor)),
        mock.patch.object(torch, "zeros", strip(real_zeros)),
        mock.patch.object(torch.Tensor, "cuda", lambda self, *a, **k: self),
        mock.patch.object(torch.cuda, "current_device", lambda: "cpu"),
    ]
    with contextlib.ExitStack() as st:
        for p in patches:
            st.enter_context(p)
        yield


# ------------------------------------------------------------------------------------------------
# fixtures
# ------------------------------------------------------------------------------------------------
def golden_cp_slice():
    """M/training/utils.py:252-350 on three synthetic layouts (incl. SURVEY.md §9's worked example)."""
    utils = importlib.import_module("long_vita_megatron.training.utils")
    cases = []
    g = torch.Generator().manual_seed(1234)

    def layout(seq, n_img, tok, start, gap):
        pos = []
        p = start
        for _ in range(n_img):
            pos.append(torch.arange(p, p + tok))
            p += tok + gap
        assert p - gap <= seq
        return torch.stack([torch.zeros(n_img, tok, dtype=torch.long), torch.stack(pos)])

    specs = [
        dict(name="survey_appendix", seq=16, cp=2, indices=torch.stack(
            [torch.zeros(2, 4, dtype=torch.long), torch.tensor([[2, 3, 4, 5], [12, 13, 14, 15]])])),
        dict(name="video_like", seq=512, cp=4, indices=layout(512, 14, 32, 1, 2)),
        dict(name="straddle", seq=1024, cp=8, indices=layout(1024, 5, 96, 30, 70)),
    ]
    for sp in specs:
        seq, cp, ind = sp["seq"], sp["cp"], sp["indices"]
        tokens = torch.randint(0, 1000, (1, seq), generator=g)
        images = torch.randn(ind.shape[1], 3, 2, 2, generator=g)
        per_rank = []
        for r in range(cp):
            STATE["cp_size"], STATE["cp_rank"] = cp, r
            a = STATE["args"]
            a.reset_position_ids, a.context_parallel_size, a.seq_length = False, cp, seq
            batch = {"tokens": tokens.clone(), "labels": tokens.clone() + 1,
                     "position_ids": torch.arange(seq)[None].clone(),
                     "external_images": images.clone(), "external_indices": ind.clone()}
            with cpu_as_cuda():
                out = utils.get_batch_on_this_cp_rank(batch)
            per_rank.append({k: v.clone() for k, v in out.items() if v is not None})
        cases.append(dict(name=sp["name"], seq=seq, cp=cp, tokens=tokens, images=images, indices=ind, out=per_rank))
    a_ = torch.tensor([7, 3, 11, 5])
    b_ = torch.tensor([1, 3, 5, 7, 9, 11])
    with cpu_as_cuda():
        ioab = utils.index_of_a_in_b(a_, b_)
    torch.save(dict(cases=cases, index_of_a_in_b=dict(a=a_, b=b_, out=ioab)), os.path.join(OUT, "cp_slice.pt"))


def golden_rope_rmsnorm():
    rpe = importlib.import_module("long_vita_megatron.core.models.common.embeddings.rotary_pos_embedding")
    te = importlib.import_module("long_vita_megatron.core.transformer.custom_layers.transformer_engine")
    utils = importlib.import_module("long_vita_megatron.training.utils")
    g = torch.Generator().manual_seed(4321)
    out = {}
    with cpu_as_cuda():
        rope = rpe.RotaryEmbedding(kv_channels=128, rotary_percent=1.0, rotary_base=1000000)
        out["inv_freq"] = rope.inv_freq.clone()
        # plain table, CP=1
        STATE["cp_size"], STATE["cp_rank"] = 1, 0
        utils.set_position_ids(None)
        emb = rope(96)
        out["emb_cp1"] = emb.clone()
        # CP=4 zig-zag slices
        out["emb_cp4"] = []
        for r in range(4):
            STATE["cp_size"], STATE["cp_rank"] = 4, r
            out["emb_cp4"].append(rope(96).clone())
        # position_ids gather (packed sequences), [s, b] as set_position_ids stores it
        STATE["cp_size"], STATE["cp_rank"] = 1, 0
        pid = torch.cat([torch.arange(40), torch.arange(56)])[:, None]
        utils.set_position_ids(pid)
        out["position_ids"] = pid
        out["emb_pid"] = rope(96).clone()
        utils.set_position_ids(None)
        # large positions (1M context) — table rows only
        big = torch.tensor([0, 1, 4095, 131071, 524288, 1048575])
        out["big_pos"] = big
        out["emb_big"] = rope(1048576)[big].clone()
        # apply, bf16 and fp32
        t = torch.randn(96, 1, 6, 128, generator=g)
        out["t"] = t
        out["apply_fp32"] = rpe.apply_rotary_pos_emb_bshd(t, emb).clone()
        out["apply_bf16"] = rpe.apply_rotary_pos_emb_bshd(t.bfloat16(), emb).clone()
        # RMSNorm (PTNorm's RMSNorm), bf16 and fp32
        x = torch.randn(33, 5120, generator=g) * 3
        w = 1 + 0.1 * torch.randn(5120, generator=g)
        n = te.RMSNorm(5120, eps=1e-6)
        n.weight.data = w.clone()
        out["rms_x"], out["rms_w"] = x, w
        out["rms_fp32"] = n(x).detach().clone()
        nb = te.RMSNorm(5120, eps=1e-6).bfloat16()
        nb.weight.data = w.bfloat16()
        out["rms_bf16"] = nb(x.bfloat16()).detach().clone()
    torch.save(out, os.path.join(OUT, "rope_rmsnorm.pt"))


def golden_embedding_scatter():
    lme = importlib.import_module("long_vita_megatron.core.models.common.embeddings.language_model_embedding")
    g = torch.Generator().manual_seed(99)
    emb = object.__new__(lme.LanguageModelEmbedding)
    torch.nn.Module.__init__(emb)
    cfg = types.SimpleNamespace(fp32_residual_connection=False, sequence_parallel=False,
                                clone_scatter_output_in_embedding=False)
    emb.config = cfg
    emb.word_embeddings = torch.nn.Embedding(50, 16)
    emb.word_embeddings.weight.data = torch.randn(50, 16, generator=g)
    emb.add_position_embedding = False
    emb.tokentype_embeddings = None
    emb.embedding_dropout = torch.nn.Identity()
    STATE["cp_size"] = 1
    ids = torch.randint(0, 50, (2, 12), generator=g)
    feats = torch.randn(3, 4, 16, generator=g)
    out = dict(weight=emb.word_embeddings.weight.data.clone(), ids=ids, feats=feats)
    with torch.no_grad():
        out["plain"] = emb(ids, None).clone()
        ind = torch.stack([torch.tensor([[0] * 4, [1] * 4, [1] * 4]),
                           torch.tensor([[1, 2, 3, 4], [0, 1, 2, 3], [8, 9, 10, 11]])])
        out["indices"] = ind
        out["with_indices"] = emb(ids, None, external_feature_dict={"features": feats, "indices": ind}).clone()
        out["with_pre_len"] = emb(ids[:1].repeat(3, 1), None,
                                  external_feature_dict={"features": feats, "pre_len": 5}).clone()
        src = torch.stack([torch.tensor([0, 0, 2, 2, 2]), torch.tensor([1, 3, 0, 1, 2])])
        tgt = torch.stack([torch.tensor([0, 0, 1, 1, 1]), torch.tensor([5, 6, 0, 1, 11])])
        out["src"], out["tgt"] = src, tgt
        out["with_src_tgt"] = emb(ids, None, external_feature_dict={"features": feats, "src_indices": src,
                                                                      "tgt_indices": tgt}).clone()
    torch.save(out, os.path.join(OUT, "embedding_scatter.pt"))


def golden_masked_linear():
    layers = importlib.import_module("long_vita_megatron.core.tensor_parallel.layers")
    g = torch.Generator().manual_seed(7)
    s, b, c, o = 24, 1, 32, 40
    x = torch.randn(s, b, c, generator=g, requires_grad=True)
    w = torch.randn(o, c, generator=g, requires_grad=True)
    mask = torch.zeros(b, s, dtype=torch.bool)
    mask[0, [3, 4, 11, 23]] = True
    fn = layers.LinearWithGradAccumulationAndAsyncCommunication
    y = fn.apply(x, w, None, False, False, False, None, mask)
    go = torch.randn(y.shape, generator=g)
    y.backward(go)
    out = dict(x=x.detach().clone(), w=w.detach().clone(), mask=mask, y=y.detach().clone(), go=go,
               dx=x.grad.clone(), dw=w.grad.clone())
    # frozen-weight path (layers.py:288-363), forward only
    wf = w.detach().clone()
    wf.requires_grad = False
    inp = x.detach()
    m = mask
    sel = torch.masked_select(inp, m.transpose(0, 1).unsqueeze(2)).reshape(-1, b, c)   # layers.py:344-348 verbatim shape logic
    out["y_frozen"] = torch.matmul(sel, wf.t())
    torch.save(out, os.path.join(OUT, "masked_linear.pt"))
    # the same function at a size the MFMA GEMM accepts (K % 64 == 0) and in the dtype the path runs in (bf16): what the HIP
    # ColumnParallelLinear module (forward + autograd backward) is compared with directly
    g = torch.Generator().manual_seed(8)
    s, b, c, o = 192, 1, 256, 320
    x = (torch.randn(s, b, c, generator=g) * 0.5).bfloat16().requires_grad_(True)
    w = (torch.randn(o, c, generator=g) * 0.05).bfloat16().requires_grad_(True)
    mask = torch.zeros(b, s, dtype=torch.bool)
    mask[0, torch.randperm(s, generator=g)[:70]] = True
    y = fn.apply(x, w, None, False, False, False, None, mask)
    go = torch.randn(y.shape, generator=g).bfloat16()
    y.backward(go)
    torch.save(dict(x=x.detach().clone(), w=w.detach().clone(), mask=mask, y=y.detach().clone(), go=go, dx=x.grad.clone(),
                    dw=w.grad.clone()), os.path.join(OUT, "masked_linear_bf16.pt"))


def golden_hf_vit():
    """The reference's HF InternVisionModel + ResamplerProjector imported UNMODIFIED
    (H/models/long_vita_qwen2_intern/modeling_intern_vit.py, resampler_projector.py), fp32 on CPU,
    seeded weights, 2 layers and the full 24 layers, one 448x448 frame."""
    import json

    import transformers  # noqa: F401  (before stubbing timm)
    timm = types.ModuleType("timm"); timm_m = types.ModuleType("timm.models"); timm_l = types.ModuleType("timm.models.layers")

    class DropPath(torch.nn.Identity):
        def __init__(self, *a, **k):
            super().__init__()

    timm_l.DropPath = DropPath
    sys.modules.update({"timm": timm, "timm.models": timm_m, "timm.models.layers": timm_l})
    # bypass long_vita/__init__.py (imports cv2): register empty parents
    for n in ["long_vita", "long_vita.models", "long_vita.models.long_vita_qwen2_intern"]:
        m = types.ModuleType(n)
        m.__path__ = [os.path.join(REF, *n.split("."))]
        sys.modules[n] = m
    civ = importlib.import_module("long_vita.models.long_vita_qwen2_intern.configuration_intern_vit")
    miv = importlib.import_module("long_vita.models.long_vita_qwen2_intern.modeling_intern_vit")
    rp = importlib.import_module("long_vita.models.long_vita_qwen2_intern.resampler_projector")
    cfg_json = json.load(open(os.path.join(REF, "long_vita/models/long_vita_qwen2_intern/config_14B.json")))

    from oracle import vit as ovit

    out = {}
    for tag, nl in (("l2", 2), ("l24", 24)):
        vcfg = dict(cfg_json["visual"]); vcfg["num_hidden_layers"] = nl; vcfg["use_flash_attn"] = False
        hcfg = civ.InternVisionConfig(**vcfg)
        model = miv.InternVisionModel(hcfg).eval().float()
        proj = rp.ResamplerProjector(types.SimpleNamespace(hidden_size=5120), hcfg).eval().float()
        ocfg = ovit.ViTConfig(num_layers=nl)
        p = ovit.init_vit_params(ocfg, seed=1234, dtype=torch.float32)
        # the bf16-rounded values are what both sides use
        p = _tree_map(p, lambda t: t.bfloat16().float())
        sd = model.state_dict()
        sd["embeddings.class_embedding"] = p["cls"]
        sd["embeddings.patch_embedding.weight"] = p["conv_w"]; sd["embeddings.patch_embedding.bias"] = p["conv_b"]
        sd["embeddings.position_embedding"] = p["pos"][None]
        for i, lp in enumerate(p["layers"]):
            pre = f"encoder.layers.{i}."
            sd[pre + "attn.qkv.weight"] = ovit.megatron_qkv_to_hf(lp["qkv_w"], 16, 64)
            sd[pre + "attn.qkv.bias"] = ovit.megatron_qkv_to_hf(lp["qkv_b"], 16, 64)
            sd[pre + "attn.proj.weight"] = lp["proj_w"]; sd[pre + "attn.proj.bias"] = lp["proj_b"]
            sd[pre + "mlp.fc1.weight"] = lp["fc1_w"]; sd[pre + "mlp.fc1.bias"] = lp["fc1_b"]
            sd[pre + "mlp.fc2.weight"] = lp["fc2_w"]; sd[pre + "mlp.fc2.bias"] = lp["fc2_b"]
            sd[pre + "norm1.weight"] = lp["ln1_w"]; sd[pre + "norm1.bias"] = lp["ln1_b"]
            sd[pre + "norm2.weight"] = lp["ln2_w"]; sd[pre + "norm2.bias"] = lp["ln2_b"]
            sd[pre + "ls1"] = lp["ls1"]; sd[pre + "ls2"] = lp["ls2"]
        model.load_state_dict(sd)
        psd = proj.state_dict()
        psd["pre_proj_layernorm.weight"] = p["proj_ln_w"]; psd["pre_proj_layernorm.bias"] = p["proj_ln_b"]
        psd["mlp.0.weight"] = p["proj_fc1"]; psd["mlp.2.weight"] = p["proj_fc2"]
        proj.load_state_dict(psd)
        g = torch.Generator().manual_seed(2024)
        images = torch.randn(1, 3, 448, 448, generator=g).bfloat16().float()
        with torch.no_grad():
            hid = model(images).last_hidden_state            # [1, 1025, 1024]
            feat = proj(hid[:, 1:, :])                       # [1, 256, 5120]   (modeling_long_vita.py:91-98)
        out[tag] = dict(hidden_sub=hid[:, ::41, ::16].clone(), feat_sub=feat[:, ::8, ::40].clone(),
                        hidden_sum=hid.double().sum(), feat_sum=feat.double().sum(),
                        hidden_absmean=hid.abs().mean(), feat_absmean=feat.abs().mean())
    out["image_seed"] = 2024
    out["weight_seed"] = 1234
    torch.save(out, os.path.join(OUT, "hf_vit.pt"))


def golden_image_processor():
    """The reference's own ImageProcessor.process_images (H/data/processor/image_processor.py:180-223) run on small
    synthetic frames (image_size 56 keeps the fixture small; the resize arithmetic does not depend on the size).
    cv2 / natsort / decord are imported at module level but not used by process_images: stubbed."""
    import types

    import numpy as np
    from PIL import Image
    for n in ("cv2", "natsort", "decord"):
        if n not in sys.modules:
            sys.modules[n] = types.ModuleType(n)
    # load the two reference files directly: the long_vita.data package __init__ pulls in torchvision / datasets
    import importlib.util

    def load(name, path):
        spec = importlib.util.spec_from_file_location(name, path)
        mod = importlib.util.module_from_spec(spec)
        sys.modules[name] = mod
        spec.loader.exec_module(mod)
        return mod
    if "long_vita" not in sys.modules:
        sys.modules["long_vita"] = types.ModuleType("long_vita")
    load("long_vita.constants", os.path.join(REF, "long_vita", "constants.py"))
    ImageProcessor = load("long_vita_image_processor",
                          os.path.join(REF, "long_vita", "data", "processor", "image_processor.py")).ImageProcessor
    rng = np.random.default_rng(77)
    out = {"cases": []}
    for norm, size, shapes in [("imagenet", 56, [(100, 37), (64, 64), (90, 160), (56, 56), (23, 41)]),
                               ("siglip", 28, [(75, 120)]), ("clip", 28, [(40, 30)])]:
        proc = ImageProcessor("", image_size=size, normalize_type=norm)
        frames = [rng.integers(0, 256, (h, w, 3), dtype=np.uint8) for h, w in shapes]
        # a smooth frame as well: bicubic over/undershoot on edges exercises clip8
        frames.append((np.indices((48, 80)).sum(0)[..., None] * np.array([3, 2, 1]) % 256).astype(np.uint8))
        ref = proc.process_images([Image.fromarray(f) for f in frames])             # [N, 3, size, size] float32
        out["cases"].append({"normalize_type": norm, "image_size": size, "frames": [torch.from_numpy(f) for f in frames],
                             "output": ref.clone(), "output_bf16": torch.tensor(ref, dtype=torch.bfloat16)})
    # dynamic tiling (--vision-process-type dynamic --max-patch-grid 12, every reference script)
    out["dynamic"] = []
    proc = ImageProcessor("dynamic", image_size=28, normalize_type="imagenet", min_patch_grid=1, max_patch_grid=12)
    for h, w in [(90, 160), (300, 70), (56, 56), (61, 200), (75, 75)]:
        frame = rng.integers(0, 256, (h, w, 3), dtype=np.uint8)
        tiles, res = proc.process_dynamic(Image.fromarray(frame))
        out["dynamic"].append({"frame": torch.from_numpy(frame), "output_bf16": torch.tensor(tiles, dtype=torch.bfloat16),
                               "resolution": tuple(int(x) for x in res)})
    # anyres tiling (process_anyres :242-266)
    out["anyres"] = []
    proc = ImageProcessor("anyres", image_size=28, normalize_type="imagenet", min_patch_grid=1, max_patch_grid=4)
    for h, w in [(90, 160), (300, 70), (56, 56), (61, 200), (40, 40)]:
        frame = rng.integers(0, 256, (h, w, 3), dtype=np.uint8)
        tiles, res = proc.process_anyres(Image.fromarray(frame))
        out["anyres"].append({"frame": torch.from_numpy(frame), "output_bf16": torch.tensor(tiles, dtype=torch.bfloat16),
                              "resolution": tuple(int(x) for x in res)})
    # frame selection rules (process_video :136-178, get_video_frames :113-134) with a fake decord / natsort
    class FakeVideoReader:
        def __init__(self, spec, num_threads=1):
            self.n, self.fps = spec

        def __len__(self):
            return self.n

        def get_avg_fps(self):
            return self.fps

        def __getitem__(self, i):
            return types.SimpleNamespace(asnumpy=lambda i=i: np.full((2, 2, 3), i % 256, dtype=np.uint8))

    mod = sys.modules["long_vita_image_processor"]
    mod.decord.VideoReader = FakeVideoReader
    import re
    mod.natsort.natsorted = lambda xs: sorted(xs, key=lambda sp: [int(t) if t.isdigit() else t for t in re.split(r"(\d+)", sp)])
    proc = ImageProcessor("", image_size=28)
    out["video_index_rule"] = []
    for (n, fps, num, mfps) in [(100, 25.0, 8, 1), (3000, 29.97, 64, 1), (40, 30.0, 8, 1), (500, 24.0, 16, 2), (7, 5.0, 8, 1)]:
        frames = proc.get_video_frames((n, fps), max_fps=mfps, num_frames=num)
        out["video_index_rule"].append(dict(total=n, fps=fps, num_frames=num, max_fps=mfps,
                                            picked=[int(np.array(f)[0, 0, 0]) for f in frames]))
    import tempfile
    out["video_dir_rule"] = []
    for tag, nfiles, maxf in [("plain", 23, 8), ("ShareGPTVideo_x", 30, 8), ("few", 3, 8)]:
        with tempfile.TemporaryDirectory() as td:
            d = os.path.join(td, tag)
            os.makedirs(d)
            for i in range(nfiles):
                Image.fromarray(np.full((4, 4, 3), i, dtype=np.uint8)).save(os.path.join(d, f"frame{i}.png"))
            _, paths = proc.process_video(d, max_num_frame=maxf, max_fps=1)
            out["video_dir_rule"].append(dict(tag=tag, nfiles=nfiles, max_num_frame=maxf,
                                              picked=[os.path.basename(x) for x in paths]))
    import PIL
    out["pillow_version"] = PIL.__version__
    torch.save(out, os.path.join(OUT, "image_processor.pt"))


class FakeTokenizer:
    """Just enough tokenizer for get_external_inputs: special tokens map to fixed ids, "\n" to 198."""

    def __init__(self):
        from long_vita import constants as c
        names = [c.IMG_TAG_TOKEN, c.VID_TAG_TOKEN, c.IMG_CONTEXT_TOKEN, c.IMG_START_TOKEN, c.IMG_END_TOKEN, c.VID_CONTEXT_TOKEN,
                 c.VID_START_TOKEN, c.VID_END_TOKEN, c.PATCH_CONTEXT_TOKEN, c.PATCH_START_TOKEN, c.PATCH_END_TOKEN]
        self.table = {n: 151700 + i for i, n in enumerate(names)}
        self.table["\n"] = 198
        self.pad_token_id, self.eos_token_id = None, 151645

    def __call__(self, text, add_special_tokens=False):
        return types.SimpleNamespace(input_ids=[self.table[text]])


class FakeProcessor:
    """Stands in for ImageProcessor: the pixel work is covered by image_processor.pt; here only the counts matter."""
    patch_size = 448

    def process_images_with_subpatch(self, spec):                             # spec = (tiles_x, tiles_y)
        tx, ty = spec
        n = tx * ty
        return torch.zeros(n + 1 if n > 1 else 1, 3, 2, 2), (tx * 448, ty * 448)

    def process_images(self, lst):
        return torch.zeros(len(lst), 3, 2, 2)

    def process_video(self, n_frames, max_num_frame, max_fps):
        return torch.zeros(min(n_frames, max_num_frame), 3, 2, 2), None


def golden_external_inputs():
    """The reference's own get_external_inputs (M/tasks/inference/module.py:493-707) under the stub Megatron, with a fake
    tokenizer / processor: token surgery for image tags (untiled, 2 x 1 and 2 x 3 tiles), video tags, both, and padding."""
    import importlib
    if "long_vita" in sys.modules and not hasattr(sys.modules["long_vita"], "__path__"):
        sys.modules["long_vita"].__path__ = [os.path.join(REF, "long_vita")]
    mod = importlib.import_module("long_vita_megatron.tasks.inference.module")
    tok = FakeTokenizer()
    from long_vita import constants as c
    IMG, VID = tok.table[c.IMG_TAG_TOKEN], tok.table[c.VID_TAG_TOKEN]
    STATE["args"].image_token_length, STATE["args"].max_num_frame, STATE["args"].max_fps, STATE["args"].bf16 = 256, 5, 1, True
    rng = torch.Generator().manual_seed(5)

    def text(n):
        return torch.randint(0, 150000, (n,), generator=rng).tolist()

    cases = [
        dict(tokens=[text(7) + [IMG] + text(5)], image_list=[(1, 1)]),
        dict(tokens=[text(3) + [IMG] + text(4) + [IMG] + text(9)], image_list=[(2, 1), (2, 3)]),
        dict(tokens=[text(10) + [VID] + text(2)], video_path_list=[4]),
        dict(tokens=[text(2) + [VID] + text(6)], video_path_list=[9]),                      # capped at max_num_frame = 5
        dict(tokens=[[IMG] + text(1)], image_path_list=[(3, 1)]),
    ]
    out = {"table": tok.table, "cases": []}
    with cpu_as_cuda():
        for c in cases:
            ext, toks, lens = mod.get_external_inputs(torch.tensor(c["tokens"]), c.get("image_list"), c.get("image_path_list"),
                                                      c.get("video_path_list"), tok, FakeProcessor())
            out["cases"].append(dict(c, out_tokens=toks.clone(), out_lengths=lens.clone(), indices=ext["indices"].clone(),
                                     n_images=int(ext["images"].shape[0]), images_dtype=str(ext["images"].dtype)))
    torch.save(out, os.path.join(OUT, "external_inputs.pt"))


DECODE_CASES = [dict(name="cp2_s32", cp=2, seq=32, prompt=5, vocab=97),            # context 5..31: crosses ctx % 8 == 0 three times
                dict(name="cp4_s64", cp=4, seq=64, prompt=13, vocab=101),
                dict(name="cp1_s24", cp=1, seq=24, prompt=3, vocab=89)]


def fake_next_token(tok: torch.Tensor, pos: torch.Tensor, vocab: int) -> torch.Tensor:
    """The "model" of the decode-loop fixture: the logits at a position are one-hot at a function of the token there and of
    its GLOBAL position, so the sampled token says which position's logits the loop picked."""
    return (tok * 7 + pos * 3 + 1) % vocab


def _decode_worker(rank: int, case: dict, port: int, out_dir: str):
    """One CP rank of the reference's decode loop (M/inference/text_generation/generation.py:33-280, with its
    get_batch_on_this_cp_rank :517-539 and sync_output :542-566) on CPU + gloo, Megatron stubbed."""
    import torch.distributed as dist
    cp, seq, vocab = case["cp"], case["seq"], case["vocab"]
    _install_stubs()
    STATE["cp_size"], STATE["cp_rank"] = cp, rank
    a = STATE["args"]
    a.use_kv_cache, a.logit_mask = False, True                       # server_cp .sh:184 (no cache under CP) + --logit-mask
    a.max_position_embeddings, a.max_tokens_to_oom, a.padded_vocab_size, a.eos_id = 1 << 20, 1 << 30, vocab, -1
    a.reset_position_ids, a.context_parallel_size, a.seq_length = False, cp, seq
    dist.init_process_group("gloo", init_method=f"tcp://127.0.0.1:{port}", rank=rank, world_size=cp)
    for mod in ("megatron.core.mpu", "megatron.core.parallel_state"):
        m = sys.modules[mod]
        m.is_pipeline_last_stage = lambda: True
        m.get_context_parallel_group = lambda: None
    sys.modules["megatron.training"].get_tokenizer = lambda: types.SimpleNamespace(eod=-1)
    gen = importlib.import_module("long_vita_megatron.inference.text_generation.generation")
    gen.copy_from_last_to_first_pipeline_stage = lambda *a_, **k_: None
    gen.broadcast_from_last_pipeline_stage = lambda size, dtype, tensor=None: tensor
    gen.broadcast_from_last_to_first_pipeline_stage = lambda size, dtype, tensor=None: tensor
    masks = []

    class FakeForwardStep:
        def __init__(self, model, batch_size, max_sequence_length, external_inputs=None):
            self.inference_params = types.SimpleNamespace(external_inputs=external_inputs, logit_mask=None, use_kv_cache=True)

        def __call__(self, tokens, position_ids, attention_mask):
            lm = self.inference_params.logit_mask
            sel = lm[0].nonzero().flatten()                          # masked_select keeps ascending position order
            masks.append(sel.tolist())
            nxt = fake_next_token(tokens[0, sel], position_ids[0, sel], vocab)
            return torch.nn.functional.one_hot(nxt, vocab).float()[None]          # [1, n_sel, V]

    gen.ForwardStep = FakeForwardStep
    g = torch.Generator().manual_seed(77)
    tokens = torch.zeros(1, seq, dtype=torch.long)
    tokens[0, :case["prompt"]] = torch.randint(0, vocab, (case["prompt"],), generator=g)
    lengths = torch.tensor([case["prompt"]])
    ext = None
    if cp > 1:                                                       # every rank must own a visual token
        c = seq // (2 * cp)
        starts = torch.arange(2 * cp) * c + 1
        ext = {"images": torch.zeros(2 * cp, 3, 2, 2),
               "indices": torch.stack([torch.zeros(2 * cp, 2, dtype=torch.long), torch.stack([starts, starts + 1], dim=1)])}
    prompt = tokens.clone()
    with cpu_as_cuda():
        for _ in gen.generate_tokens_probs_and_return_on_first_stage(None, tokens, lengths, external_inputs=ext):
            pass
    torch.save(dict(masks=masks, tokens=tokens, prompt=prompt, lengths=lengths), os.path.join(out_dir, f"rank{rank}.pt"))
    dist.barrier()
    dist.destroy_process_group()


def golden_decode_loop():
    """The reference's own decode loop on CP = 1 / 2 / 4 gloo ranks with a position-revealing fake model: per step the
    logit-mask positions of every rank, and the tokens it generates (which encode the block sync_output + the picker chose)."""
    out = {"cases": []}
    for case in DECODE_CASES:
        ranks = _run_ranks(_decode_worker, case)
        for r_ in ranks[1:]:
            assert torch.equal(r_["tokens"], ranks[0]["tokens"])     # every rank generated the same text
        out["cases"].append(dict(case, prompt=ranks[0]["prompt"], lengths=ranks[0]["lengths"], tokens=ranks[0]["tokens"],
                                 masks=[r_["masks"] for r_ in ranks]))
    torch.save(out, os.path.join(OUT, "decode_loop.pt"))


ATTN_CASES = [dict(name="llm_gqa_causal", sq=48, b=1, np=8, ng=2, hn=16, causal=True),
              dict(name="vit_mha_full", sq=17, b=2, np=4, ng=4, hn=8, causal=False)]


def attn_case_inputs(case: dict):
    g = torch.Generator().manual_seed(4242 + case["sq"])
    q = torch.randn(case["sq"], case["b"], case["np"], case["hn"], generator=g)
    k = torch.randn(case["sq"], case["b"], case["ng"], case["hn"], generator=g)
    v = torch.randn(case["sq"], case["b"], case["ng"], case["hn"], generator=g)
    return q, k, v


def golden_unfused_attention():
    """The reference's unfused core attention (M/core/transformer/dot_product_attention.py:151-291: GQA repeat_interleave,
    baddbmm with 1/norm_factor, softmax, bmm, [sq, b, hp] layout) run with a stand-in `self`.  The one external piece is
    Megatron's `scale_mask_softmax`; its torch path (Megatron-LM core_r0.7.0 megatron/core/fusions/fused_softmax.py,
    forward_torch_softmax with attention_mask_func: masked_fill(mask, -10000.0), softmax over the last dim) is restated here."""
    dpa = importlib.import_module("long_vita_megatron.core.transformer.dot_product_attention")
    STATE["args"].use_flash_attn = False
    dpa.parallel_state = types.SimpleNamespace(get_global_memory_buffer=lambda: types.SimpleNamespace(
        get_tensor=lambda shape, dtype, name: torch.empty(shape, dtype=dtype)))

    def scale_mask_softmax(scores, mask):
        if mask is not None:
            scores = scores.masked_fill(mask, -10000.0)
        return torch.nn.Softmax(dim=-1)(scores)

    fwd = dpa.dot_product_attention_forward_wrapper(lambda *a_, **k_: None)
    out = {"cases": []}
    for case in ATTN_CASES:
        q, k, v = attn_case_inputs(case)
        me = types.SimpleNamespace(num_attention_heads_per_partition=case["np"], num_query_groups_per_partition=case["ng"],
                                   alibi=None, norm_factor=case["hn"] ** 0.5, attn_logit_softcapping=None, square_alibi_mask=False,
                                   scale_mask_softmax=scale_mask_softmax, attention_dropout=lambda x: x,
                                   config=types.SimpleNamespace(sequence_parallel=False),
                                   hidden_size_per_partition=case["np"] * case["hn"])
        mask = torch.triu(torch.ones(case["sq"], case["sq"], dtype=torch.bool), 1)[None, None] if case["causal"] else None
        ctx = fwd(me, q.clone(), k.clone(), v.clone(), mask, None, None)
        out["cases"].append(dict(case, out=ctx.clone()))
    torch.save(out, os.path.join(OUT, "unfused_attention.pt"))


def golden_unfused_attention_bf16():
    """The same reference wrapper run as Megatron runs it in half precision (`--bf16`): bf16 q / k / v, so `torch.baddbmm` writes
    bf16 scores, scale_mask_softmax (core_r0.7.0 fused_softmax.py forward_torch_softmax, restated: `input.float()` when
    input_in_float16 and softmax_in_fp32 — `--attention-softmax-in-fp32` — then masked_fill(-10000), softmax,
    `probs.bfloat16()`) hands bf16 probabilities to `torch.bmm`, which writes a bf16 context — the dtype chain that
    oracle.attention.core_attention(chain=True) restates; plus torch autograd through it (bf16 gradients)."""
    dpa = importlib.import_module("long_vita_megatron.core.transformer.dot_product_attention")
    STATE["args"].use_flash_attn = False
    dpa.parallel_state = types.SimpleNamespace(get_global_memory_buffer=lambda: types.SimpleNamespace(
        get_tensor=lambda shape, dtype, name: torch.empty(shape, dtype=dtype)))

    def scale_mask_softmax(scores, mask):
        assert scores.dtype == torch.bfloat16
        x = scores.float()
        if mask is not None:
            x = x.masked_fill(mask, -10000.0)
        return torch.nn.Softmax(dim=-1)(x).bfloat16()

    fwd = dpa.dot_product_attention_forward_wrapper(lambda *a_, **k_: None)
    out = {"cases": []}
    for case in ATTN_CASES_BF16:
        q, k, v = (t.bfloat16().requires_grad_(True) for t in attn_case_inputs(case))
        me = types.SimpleNamespace(num_attention_heads_per_partition=case["np"], num_query_groups_per_partition=case["ng"],
                                   alibi=None, norm_factor=case["hn"] ** 0.5, attn_logit_softcapping=None, square_alibi_mask=False,
                                   scale_mask_softmax=scale_mask_softmax, attention_dropout=lambda x: x,
                                   config=types.SimpleNamespace(sequence_parallel=False),
                                   hidden_size_per_partition=case["np"] * case["hn"])
        mask = torch.triu(torch.ones(case["sq"], case["sq"], dtype=torch.bool), 1)[None, None] if case["causal"] else None
        ctx = fwd(me, q, k, v, mask, None, None)
        assert ctx.dtype == torch.bfloat16
        go = torch.randn(ctx.shape, generator=torch.Generator().manual_seed(77 + case["sq"])).bfloat16()
        ctx.backward(go)
        out["cases"].append(dict(case, out=ctx.detach().clone(), go=go, dq=q.grad.clone(), dk=k.grad.clone(), dv=v.grad.clone()))
    torch.save(out, os.path.join(OUT, "unfused_attention_bf16.pt"))


ATTN_CASES_BF16 = [dict(name="llm_gqa_causal_bf16", sq=320, b=1, np=10, ng=2, hn=128, causal=True),
                   dict(name="vit_mha_full_bf16", sq=65, b=2, np=4, ng=4, hn=64, causal=False)]


VIT_CONVERT_SHAPES = {        # one InternViT-300M layer at full width (the converter hard-codes 16 heads x 64, hidden 1024)
    "embeddings.class_embedding": (1, 1, 1024), "embeddings.position_embedding": (1, 1025, 1024),
    "embeddings.patch_embedding.weight": (1024, 3, 14, 14), "embeddings.patch_embedding.bias": (1024,),
    "encoder.layers.0.attn.qkv.weight": (3072, 1024), "encoder.layers.0.attn.qkv.bias": (3072,),
    "encoder.layers.0.attn.proj.weight": (1024, 1024), "encoder.layers.0.attn.proj.bias": (1024,),
    "encoder.layers.0.norm1.weight": (1024,), "encoder.layers.0.norm1.bias": (1024,),
    "encoder.layers.0.mlp.fc1.weight": (4096, 1024), "encoder.layers.0.mlp.fc1.bias": (4096,),
    "encoder.layers.0.mlp.fc2.weight": (1024, 4096), "encoder.layers.0.mlp.fc2.bias": (1024,),
    "encoder.layers.0.norm2.weight": (1024,), "encoder.layers.0.norm2.bias": (1024,),
    "encoder.layers.0.ls1": (1024,), "encoder.layers.0.ls2": (1024,)}
LLM_CONVERT_DIMS = dict(hidden=64, heads=8, groups=2, head_dim=8, ffn=96, vocab=40, layers=2)


def llm_convert_hf_state(seed: int = 31):
    """A small Qwen2-shaped transformers state dict (names of Qwen2ForCausalLM)."""
    d = LLM_CONVERT_DIMS
    g = torch.Generator().manual_seed(seed)
    r = lambda *shape: torch.randn(*shape, generator=g)
    sd = {"model.embed_tokens.weight": r(d["vocab"], d["hidden"]), "model.norm.weight": r(d["hidden"]),
          "lm_head.weight": r(d["vocab"], d["hidden"])}
    for i in range(d["layers"]):
        pre = f"model.layers.{i}."
        sd.update({pre + "self_attn.q_proj.weight": r(d["heads"] * d["head_dim"], d["hidden"]),
                   pre + "self_attn.q_proj.bias": r(d["heads"] * d["head_dim"]),
                   pre + "self_attn.k_proj.weight": r(d["groups"] * d["head_dim"], d["hidden"]),
                   pre + "self_attn.k_proj.bias": r(d["groups"] * d["head_dim"]),
                   pre + "self_attn.v_proj.weight": r(d["groups"] * d["head_dim"], d["hidden"]),
                   pre + "self_attn.v_proj.bias": r(d["groups"] * d["head_dim"]),
                   pre + "self_attn.o_proj.weight": r(d["hidden"], d["heads"] * d["head_dim"]),
                   pre + "mlp.gate_proj.weight": r(d["ffn"], d["hidden"]), pre + "mlp.up_proj.weight": r(d["ffn"], d["hidden"]),
                   pre + "mlp.down_proj.weight": r(d["hidden"], d["ffn"]),
                   pre + "input_layernorm.weight": r(d["hidden"]), pre + "post_attention_layernorm.weight": r(d["hidden"])})
    return sd


def golden_converters():
    """(A) L/ckpt_converter_intern_vit.py:convert run on a marker state dict (every element encodes its tensor, row and
    column), tensor-parallel size 2, --use-te: the fixture keeps, per written tensor and rank, which source rows / columns it
    holds.  (B) R/tools/hf2mcore_long_vita.py:convert_checkpoint_from_transformers_to_megatron (+ safe_copy), source executed
    on stand-in module trees: the Megatron-side language-model state dict it fills."""
    import ast
    import tempfile

    # ---- (A) InternViT ------------------------------------------------------------------------------------------------
    conv = importlib.import_module("long_vita_modellink.ckpt_converter_intern_vit")
    names = list(VIT_CONVERT_SHAPES)
    sd = {}
    def two_d(shape):                       # leading singleton dims dropped; [rows, everything else]
        shape = list(shape)
        while len(shape) > 1 and shape[0] == 1:
            shape.pop(0)
        cols = 1
        for x in shape[1:]:
            cols *= x
        return shape[0], cols

    for tid, name in enumerate(names):
        shape = VIT_CONVERT_SHAPES[name]
        rows, cols = two_d(shape)
        assert rows < 65536 and cols < 65536
        mark = (tid << 32) + (torch.arange(rows, dtype=torch.float64)[:, None] * 65536.0) + torch.arange(cols, dtype=torch.float64)[None]
        sd[name] = mark.reshape(shape)
    fake = types.SimpleNamespace(state_dict=lambda: sd)
    fake.cpu = lambda: fake
    fake.eval = lambda: fake
    import transformers
    vit = {"names": names, "ranks": []}
    with tempfile.TemporaryDirectory() as d, mock.patch.object(transformers.AutoModel, "from_pretrained", lambda *a_, **k_: fake), \
            contextlib.redirect_stdout(open(os.devnull, "w")):
        conv.convert("unused", d, 2, True)
        assert open(os.path.join(d, "latest_checkpointed_iteration.txt")).read() == "1"
        for i in range(2):
            shard = torch.load(os.path.join(d, "iter_0000001", f"mp_rank_0{i}", "model_optim_rng.pt"), weights_only=False)["model"]
            entry = {}
            for name, t in shard.items():
                if t is None:
                    entry[name] = None                                  # TE _extra_state placeholders
                    continue
                flat = t.reshape(*two_d(t.shape))
                v = flat.to(torch.int64)
                tid = int(v[0, 0] >> 32)
                rowmap = ((v[:, 0] >> 16) & 0xFFFF).tolist()      # the tensor is src[rowmap][:, colmap] of ONE source tensor
                colmap = (v[0, :] & 0xFFFF).tolist()
                assert torch.equal(v, (tid << 32) + (torch.tensor(rowmap)[:, None] << 16) + torch.tensor(colmap)[None])
                entry[name] = dict(src=names[tid], shape=tuple(t.shape), rowmap=rowmap, colmap=colmap)
            vit["ranks"].append(entry)

    # ---- (B) Qwen2 LLM ------------------------------------------------------------------------------------------------
    path = os.path.join(REF, "tools", "hf2mcore_long_vita.py")
    src = open(path).read()
    tree = ast.parse(src)
    ns = {"torch": torch}
    for fn in tree.body:
        if isinstance(fn, ast.FunctionDef) and fn.name in ("safe_copy", "convert_checkpoint_from_transformers_to_megatron"):
            exec(compile(ast.get_source_segment(src, fn), path, "exec"), ns)
    dm = LLM_CONVERT_DIMS
    hf_sd = llm_convert_hf_state()
    P = lambda *shape: torch.nn.Parameter(torch.zeros(*shape))

    class Holder(torch.nn.Module):
        def __init__(self, **kw):
            super().__init__()
            for k_, v_ in kw.items():
                setattr(self, k_, v_)

    def lin(w, b=None):
        h = Holder(weight=torch.nn.Parameter(w.clone()))
        if b is not None:
            h.bias = torch.nn.Parameter(b.clone())
        return h

    hf_layers = torch.nn.ModuleList()
    for i in range(dm["layers"]):
        pre = f"model.layers.{i}."
        hf_layers.append(Holder(
            self_attn=Holder(q_proj=lin(hf_sd[pre + "self_attn.q_proj.weight"], hf_sd[pre + "self_attn.q_proj.bias"]),
                             k_proj=lin(hf_sd[pre + "self_attn.k_proj.weight"], hf_sd[pre + "self_attn.k_proj.bias"]),
                             v_proj=lin(hf_sd[pre + "self_attn.v_proj.weight"], hf_sd[pre + "self_attn.v_proj.bias"]),
                             o_proj=lin(hf_sd[pre + "self_attn.o_proj.weight"])),
            mlp=Holder(gate_proj=lin(hf_sd[pre + "mlp.gate_proj.weight"]), up_proj=lin(hf_sd[pre + "mlp.up_proj.weight"]),
                       down_proj=lin(hf_sd[pre + "mlp.down_proj.weight"])),
            input_layernorm=lin(hf_sd[pre + "input_layernorm.weight"]),
            post_attention_layernorm=lin(hf_sd[pre + "post_attention_layernorm.weight"])))
    # the vision half of the function (:533-587) is for another ViT family: give it an empty one
    vis = lambda: Holder(rotary_pos_emb=Holder(inv_freq=torch.zeros(2)), patch_embed=Holder(proj=lin(torch.zeros(2, 2))),
                         blocks=torch.nn.ModuleList(), merger=Holder(ln_q=lin(torch.zeros(2), torch.zeros(2)),
                                                                     mlp=torch.nn.ModuleList([lin(torch.zeros(2, 2), torch.zeros(2)), Holder(),
                                                                                              lin(torch.zeros(2, 2), torch.zeros(2))])))
    hfmodel = Holder(visual=vis(), model=Holder(embed_tokens=lin(hf_sd["model.embed_tokens.weight"]), layers=hf_layers,
                                                norm=lin(hf_sd["model.norm.weight"])), lm_head=lin(hf_sd["lm_head.weight"]))
    qkv_out = (dm["heads"] + 2 * dm["groups"]) * dm["head_dim"]
    mg_layers = torch.nn.ModuleList([Holder(
        self_attention=Holder(linear_qkv=Holder(layer_norm_weight=P(dm["hidden"]), weight=P(qkv_out, dm["hidden"]), bias=P(qkv_out)),
                              linear_proj=Holder(weight=P(dm["hidden"], dm["heads"] * dm["head_dim"]))),
        mlp=Holder(linear_fc1=Holder(layer_norm_weight=P(dm["hidden"]), weight=P(2 * dm["ffn"], dm["hidden"])),
                   linear_fc2=Holder(weight=P(dm["hidden"], dm["ffn"])))) for _ in range(dm["layers"])])
    mgvis = Holder(rotary_pos_emb=Holder(inv_freq=torch.zeros(2)), patch_embed=Holder(proj=lin(torch.zeros(2, 2))),
                   decoder=Holder(layers=torch.nn.ModuleList(), final_layernorm=lin(torch.zeros(2), torch.zeros(2))),
                   projection=Holder(encoder=Holder(linear_fc1=lin(torch.zeros(2, 2), torch.zeros(2)),
                                                    linear_fc2=lin(torch.zeros(2, 2), torch.zeros(2)))))
    mgvis.config = types.SimpleNamespace(hidden_size=2, num_query_groups=1, num_attention_heads=1)
    mgmodel = Holder(vision_model=mgvis, language_model=Holder(
        embedding=Holder(word_embeddings=Holder(weight=P(dm["vocab"], dm["hidden"]))),
        decoder=Holder(layers=mg_layers, final_layernorm=Holder(weight=P(dm["hidden"]))),
        output_layer=Holder(weight=P(dm["vocab"], dm["hidden"]))))
    args = types.SimpleNamespace(fp16=False, bf16=False, num_attention_heads=dm["heads"], num_query_groups=dm["groups"],
                                 hidden_size=dm["hidden"], untie_embeddings_and_output_weights=True)
    ns["convert_checkpoint_from_transformers_to_megatron"](hfmodel, mgmodel, args)
    mg_sd = {k_: v_.detach().clone() for k_, v_ in mgmodel.language_model.state_dict().items()}
    assert all(float(v_.abs().sum()) > 0 for v_ in mg_sd.values())                 # every Megatron tensor was filled
    torch.save(dict(vit=vit, llm=dict(dims=dm, megatron_state=mg_sd)), os.path.join(OUT, "converters.pt"))


SAMPLING_CASES = [dict(top_k=0, top_p=0.0, temperature=1.0), dict(top_k=1, top_p=0.0, temperature=1.0),
                  dict(top_k=5, top_p=0.0, temperature=0.7), dict(top_k=0, top_p=0.8, temperature=1.3),
                  dict(top_k=12, top_p=0.5, temperature=1.0), dict(top_k=0, top_p=1.0, temperature=1.0),
                  dict(top_k=3, top_p=0.0, temperature=1.0, ties=True)]


def sampling_case_logits(i: int, case: dict):
    g = torch.Generator().manual_seed(900 + i)
    logits = torch.randn(3, 41, generator=g) * 2.5
    if case.get("ties"):
        logits[:, 5] = logits[:, 9] = logits.max(dim=-1).values - 0.5       # two equal candidates around the k-th value
    return logits


def golden_sampling():
    """_sample_strategy + top_k_logits (M/inference/text_generation/generation.py:473-512), source executed: the filtered
    distribution and, under a fixed torch seed, the sampled token; plus the greedy branch."""
    import ast
    import torch.nn.functional as F
    path = os.path.join(REF, "long_vita_megatron", "inference", "text_generation", "generation.py")
    src = open(path).read()
    ns = {"torch": torch, "F": F}
    for fn in ast.parse(src).body:
        if isinstance(fn, ast.FunctionDef) and fn.name in ("_sample_strategy", "top_k_logits"):
            exec(compile(ast.get_source_segment(src, fn), path, "exec"), ns)
    out = []
    for i, case in enumerate(SAMPLING_CASES):
        logits = sampling_case_logits(i, case).bfloat16()            # the model hands over bf16 logits
        torch.manual_seed(5 + i)
        probs, tok = ns["_sample_strategy"](logits.clone(), True, top_k=case["top_k"], top_p=case["top_p"],
                                            temperature=case["temperature"])
        _, greedy = ns["_sample_strategy"](logits.clone(), False)
        out.append(dict(case, probs=probs.clone(), token=tok.clone(), greedy=greedy.clone()))
    torch.save(dict(cases=out), os.path.join(OUT, "sampling.pt"))


def patch_scenario(manager, Patch, tag: str):
    """One script against a patch manager (the reference's M/patch_utils.py or the mirror): returns the observable outcomes."""
    out = []

    def rec(label, fn):
        try:
            out.append((label, repr(fn())))
        except Exception as e:  # noqa: BLE001
            out.append((label, f"{type(e).__name__}: {e}".replace(tag, "TAG").replace("oracle.make_golden.", "").replace("__main__.", "")))

    manager.patches_info = {}
    T, H = f"{tag}_target", f"{tag}_holder"
    mod, holder = types.ModuleType(T), types.ModuleType(H)
    mod.f = lambda x: x + 1
    mod.g = lambda x: x + 2
    mod.h = lambda x: x + 3
    mod.nothing = None

    class K:
        def m(self, x):
            return x * 2
    mod.K = K
    holder.f, holder.g = mod.f, mod.g                                  # `from target import f, g` made before patching
    holder.brand_new = None                                            # same name, value None: id(None) matches (:65-70)
    sys.modules[T], sys.modules[H] = mod, holder

    def add_wrapper(fn):
        return lambda x: fn(x) + 1000

    def times_decorator(fn):
        return lambda x: fn(x) * 3

    def method_wrapper(fn):
        return lambda self, x: fn(self, x) * 10

    rec("register f", lambda: manager.register_patch(f"{T}.f", lambda x: x + 100))
    rec("second outright f", lambda: manager.register_patch(f"{T}.f", lambda x: x))
    rec("forced f", lambda: manager.register_patch(f"{T}.f", lambda x: x + 200, force_patch=True))
    rec("wrapper on replaced f", lambda: manager.register_patch(f"{T}.f", add_wrapper))
    rec("two wrappers g", lambda: (manager.register_patch(f"{T}.g", add_wrapper), manager.register_patch(f"{T}.g", times_decorator)))
    rec("class attr", lambda: manager.register_patch(f"{T}.K.m", method_wrapper))
    rec("dummy pkg", lambda: manager.register_patch(f"{tag}_missing.sub.fn", None, create_dummy=True))
    rec("missing attr of module", lambda: manager.register_patch(f"{T}.brand_new", lambda: "made"))
    rec("None on existing h", lambda: manager.register_patch(f"{T}.h", None))
    rec("apply", lambda: manager.apply_patches())
    rec("f", lambda: mod.f(1))
    rec("holder.f", lambda: holder.f(1))
    rec("g", lambda: mod.g(1))
    rec("holder.g", lambda: holder.g(1))
    rec("K.m", lambda: mod.K().m(3))
    rec("dummy call", lambda: sys.modules[f"{tag}_missing.sub"].fn())
    rec("dummy module file", lambda: sys.modules[f"{tag}_missing.sub"].__file__)
    rec("brand_new", lambda: mod.brand_new())
    rec("holder.brand_new", lambda: holder.brand_new())
    rec("h", lambda: mod.h(1))
    rec("apply again", lambda: manager.apply_patches())
    rec("f after second apply", lambda: mod.f(1))
    rec("re-register wrapper f", lambda: manager.register_patch(f"{T}.f", add_wrapper))
    rec("apply third", lambda: manager.apply_patches())
    rec("f after re-register", lambda: mod.f(1))
    rec("missing module", lambda: Patch(f"{tag}_nowhere.x", lambda: 0, False).apply_patch())
    rec("missing class attr", lambda: Patch(f"{T}.K.absent", lambda: 0, False).apply_patch())
    rec("missing class attr dummy", lambda: Patch(f"{T}.K.absent2", None, True).apply_patch())
    rec("absent2 call", lambda: mod.K.absent2())
    rec("bare module", lambda: Patch(T, lambda: 0, False).apply_patch())
    for k in [k for k in sys.modules if k.startswith(tag)]:
        sys.modules.pop(k)
    manager.patches_info = {}
    return out


def _patch_manager_inproc():
    import importlib.util
    spec = importlib.util.spec_from_file_location("vita_ref_patch_utils", os.path.join(REF, "long_vita_megatron", "patch_utils.py"))
    ref = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(ref)
    torch.save(dict(outcomes=patch_scenario(ref.MindSpeedPatchesManager, ref.Patch, "vitaref")), os.path.join(OUT, "patch_manager.pt"))


def golden_patch_manager():
    """M/patch_utils.py (stdlib only) loaded from its file and driven through patch_scenario — in a fresh interpreter: the
    reference probes EVERY imported module with hasattr (:65-70), and on transformers' lazy modules that probe imports
    optional dependencies (torchvision ...) and raises, which is a property of the surrounding process, not of the manager."""
    import subprocess
    subprocess.run([sys.executable, "-m", "oracle.make_golden", "_patch_manager_inproc"], check=True,
                   cwd=os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def golden_adaptor_targets():
    """The plugin surface as the reference spells it: every `aspm.register_patch('<dotted target>', <replacement>)` call
    site of M/megatron_adaptor.py (read with ast — importing the file would run it against a real Megatron), grouped by the
    function that makes it, plus the functions exe_adaptation actually calls, and the direct attribute assignments."""
    import ast
    path = os.path.join(REF, "long_vita_megatron", "megatron_adaptor.py")
    src = open(path).read()
    tree = ast.parse(src)
    groups, assigns = {}, []
    for fn in [n_ for n_ in tree.body if isinstance(n_, ast.FunctionDef)]:
        calls = []
        for node in ast.walk(fn):
            if (isinstance(node, ast.Call) and isinstance(node.func, ast.Attribute) and node.func.attr == "register_patch"
                    and node.args and isinstance(node.args[0], ast.Constant) and isinstance(node.args[0].value, str)):
                calls.append((node.lineno, node.args[0].value, ast.unparse(node.args[1]) if len(node.args) > 1 else None,
                              {k_.arg: ast.unparse(k_.value) for k_ in node.keywords}))
            if isinstance(node, ast.Assign) and isinstance(node.targets[0], ast.Attribute) and not isinstance(node.value, ast.Constant):
                tgt = ast.unparse(node.targets[0])
                if tgt.startswith("megatron."):
                    assigns.append((node.lineno, tgt, ast.unparse(node.value)))
        groups[fn.name] = [c[1:] for c in sorted(calls)]
    exe = next(n_ for n_ in tree.body if isinstance(n_, ast.FunctionDef) and n_.name == "exe_adaptation")
    called = [n_.func.id for n_ in ast.walk(exe) if isinstance(n_, ast.Call) and isinstance(n_.func, ast.Name) and n_.func.id in groups]
    called = [c for c in groups if c in called]                       # source order of the definitions
    live = [t for c in called + ["exe_adaptation"] for t in groups[c]]
    torch.save(dict(groups=groups, called=called, live_targets=live, assignments=[a_[1:] for a_ in sorted(assigns)]),
               os.path.join(OUT, "adaptor_targets.pt"))


def golden_packed_positions():
    """--reset-position-ids --reset-attention-mask (stage-2 packing): M/training/utils.py:get_ltor_masks_and_position_ids
    (:192-250) on token rows with EOD tokens -> the position ids with resets and the block-diagonal causal mask;
    compute_actual_seq_len (:53-57) on those position ids."""
    utils = importlib.import_module("long_vita_megatron.training.utils")
    EOD = 3
    g = torch.Generator().manual_seed(8)
    rows = []
    for seq, eods in [(24, [5, 6, 17]), (32, [0, 9, 31]), (16, []), (40, [7, 8, 9, 30])]:
        data = torch.randint(4, 100, (1, seq), generator=g)
        data[0, eods] = EOD
        mask, loss_mask, pos = utils.get_ltor_masks_and_position_ids(data, EOD, True, True, True)
        rows.append(dict(data=data, eod=EOD, attention_mask=mask.clone(), loss_mask=loss_mask.clone(), position_ids=pos.clone(),
                         actual_seq_len=utils.compute_actual_seq_len(pos[0])))
    torch.save(dict(rows=rows), os.path.join(OUT, "packed_positions.pt"))


CKPT_SCRIPT_KEYS = ["embedding.word_embeddings.weight", "decoder.layers.0.input_layernorm.weight",
                    "decoder.layers.0.self_attention.linear_qkv.weight", "decoder.layers.0.pre_mlp_layernorm.weight",
                    "decoder.layers.0.mlp.linear_fc1.weight", "decoder.final_layernorm.weight", "output_layer.weight",
                    "unused_parameter", "external_feature_model.vit.decoder.layers.0.input_layernorm.weight",
                    "external_feature_model.vit.conv1.weight", "external_feature_model.projection.encoder.linear_fc1.weight"]


def golden_ckpt_scripts():
    """M/ckpt_convert_modellink_to_megatron_with_te.py:convert (imported) and M/ckpt_split_llm_and_vit.py (a script with
    hard-coded paths: its three path constants are re-pointed, the rest of its source runs as is) on a two-rank checkpoint
    directory whose tensors are their own key index: which keys / values end up where."""
    import ast
    import tempfile
    conv = importlib.import_module("long_vita_megatron.ckpt_convert_modellink_to_megatron_with_te")

    def write(root, it=7):
        for r in range(2):
            d = os.path.join(root, f"iter_{it:07d}", f"mp_rank_{r:02d}_000")
            os.makedirs(d)
            torch.save({"model": {k: torch.tensor([100 * r + i]) for i, k in enumerate(CKPT_SCRIPT_KEYS)}, "iteration": it},
                       os.path.join(d, "model_optim_rng.pt"))
        open(os.path.join(root, "latest_checkpointed_iteration.txt"), "w").write(str(it))

    def read(root):
        out = {}
        for base, _dirs, files in os.walk(root):
            for f in files:
                if f.endswith(".pt"):
                    sd = torch.load(os.path.join(base, f), weights_only=False)["model"]
                    out[os.path.relpath(os.path.join(base, f), root)] = {k: int(v) for k, v in sd.items()}
        return out

    with tempfile.TemporaryDirectory() as d, contextlib.redirect_stdout(open(os.devnull, "w")):
        src_dir, te_dir = os.path.join(d, "in"), os.path.join(d, "te")
        write(src_dir)
        conv.convert(src_dir, te_dir)
        renamed = read(te_dir)
        tracker = open(os.path.join(te_dir, "latest_checkpointed_iteration.txt")).read()
        path = os.path.join(REF, "long_vita_megatron", "ckpt_split_llm_and_vit.py")
        tree = ast.parse(open(path).read())
        load = os.path.join(src_dir, "iter_0000007") + "/"
        consts = {"CKPT_LOAD_DIR": load, "LLM_SAVE_DIR": os.path.join(d, "llm", "iter_0000007") + "/",
                  "VIT_SAVE_DIR": os.path.join(d, "vit", "iter_0000007") + "/"}
        for node in tree.body:
            if isinstance(node, ast.Assign) and node.targets[0].id in consts:
                node.value = ast.Constant(consts[node.targets[0].id])
        exec(compile(ast.fix_missing_locations(tree), path, "exec"), {"__name__": "split"})
        llm, vit = read(os.path.join(d, "llm")), read(os.path.join(d, "vit"))
    torch.save(dict(keys=CKPT_SCRIPT_KEYS, renamed=renamed, tracker=tracker, llm=llm, vit=vit), os.path.join(OUT, "ckpt_scripts.pt"))


LOSS_CASES = [dict(name="cp1_instruction", cp=1, instruction=True, n=[9], ones_mask=False),
              dict(name="cp2_logit_mask", cp=2, instruction=True, n=[6, 11], ones_mask=True),      # forward_step :866-867
              dict(name="cp4_plain", cp=4, instruction=False, n=[5, 8, 3, 7], ones_mask=False)]


def loss_case_inputs(case: dict, rank: int):
    """Per-token losses [1, n_r] and the loss mask of one rank ([1, n_r + 1] when the instruction shift applies)."""
    g = torch.Generator().manual_seed(1000 + 17 * rank + len(case["name"]))
    n = case["n"][rank]
    losses = torch.rand(1, n, generator=g) * 9.0
    width = n + 1 if case["instruction"] else n
    mask = torch.ones(1, width) if case["ones_mask"] else (torch.rand(1, width, generator=g) > 0.4).float()
    return losses, mask


def _loss_worker(rank: int, case: dict, port: int, out_dir: str):
    """The reference's loss_func (M/pretrain_long_vita.py:778-838) — its source text is taken from the file and executed as
    is (the module itself cannot be imported: it pulls the whole Megatron / data stack) on gloo ranks."""
    import ast

    import torch.distributed as dist
    cp = case["cp"]
    dist.init_process_group("gloo", init_method=f"tcp://127.0.0.1:{port}", rank=rank, world_size=cp)
    dp_group = None
    for r in range(cp):                                            # DP = 1: every rank is its own data-parallel group
        grp = dist.new_group([r])
        if r == rank:
            dp_group = grp
    path = os.path.join(REF, "long_vita_megatron", "pretrain_long_vita.py")
    src = open(path).read()
    fn = next(n_ for n_ in ast.parse(src).body if isinstance(n_, ast.FunctionDef) and n_.name == "loss_func")
    args = types.SimpleNamespace(is_instruction_dataset=case["instruction"], context_parallel_size=cp, save=out_dir,
                                 check_for_nan_in_loss_and_grad=True)
    mpu = types.SimpleNamespace(get_context_parallel_group=lambda: None, get_data_parallel_group=lambda: dp_group)
    ns = {"torch": torch, "os": os, "get_args": lambda: args, "mpu": mpu, "LOSS_PRINT_ONCE": True}
    exec(compile(ast.get_source_segment(src, fn), path, "exec"), ns)
    losses, mask = loss_case_inputs(case, rank)
    with cpu_as_cuda():
        loss, ntok, report = ns["loss_func"](mask.clone(), losses.clone())
    torch.save(dict(loss=loss, num_tokens=ntok, report=[x.clone() for x in report["lm loss"]]), os.path.join(out_dir, f"rank{rank}.pt"))
    dist.barrier()
    dist.destroy_process_group()


def _run_ranks(worker, case):
    import socket
    import tempfile

    import torch.multiprocessing as mp
    with socket.socket() as so:
        so.bind(("127.0.0.1", 0))
        port = so.getsockname()[1]
    with tempfile.TemporaryDirectory() as d:
        ctx = mp.get_context("spawn")
        procs = [ctx.Process(target=worker, args=(r, case, port, d)) for r in range(case["cp"])]
        [p_.start() for p_ in procs]
        [p_.join(600) for p_ in procs]
        assert all(p_.exitcode == 0 for p_ in procs), [p_.exitcode for p_ in procs]
        return [torch.load(os.path.join(d, f"rank{r}.pt")) for r in range(case["cp"])]


TP_BATCH_CASES = [dict(name="video", cp=2, images=True, reset=False), dict(name="text_only", cp=2, images=False, reset=False),
                  dict(name="packed", cp=2, images=True, reset=True)]


def tp_batch_data(case):
    """What the dataloader hands tensor-parallel rank 0 (H/data/dataset_qwen2.py item keys), preceded by an item without tokens."""
    g = torch.Generator().manual_seed(len(case["name"]))
    b, s = 2, 64
    d = {"tokens": torch.randint(0, 1000, (b, s), generator=g), "labels": torch.randint(0, 1000, (b, s), generator=g),
         "loss_mask": (torch.rand(b, s, generator=g) > 0.5).float(), "position_ids": torch.arange(s).repeat(b, 1)}
    if case["images"]:
        d["images"] = torch.randn(3, 3, 28, 28, generator=g)
        d["image_indices"] = torch.stack([torch.tensor([[0] * 4, [1] * 4, [1] * 4]), torch.randint(0, s, (3, 4), generator=g)]).to(torch.int32)
    if case["reset"]:
        d["actual_seq_len"] = torch.tensor([10, 40, 64, 74, 128])
    return [{"not_a_batch": torch.zeros(1)}, d]


def _tp_batch_worker(rank: int, case: dict, port: int, out_dir: str):
    """One tensor-parallel rank of the reference's get_batch_on_this_tp_rank (M/training/utils.py:410-626) on CPU + gloo:
    `.cuda()` / current_device re-pointed at the host, Megatron's mpu stubbed with TP = 2."""
    import torch.distributed as dist
    _install_stubs()
    a = STATE["args"]
    a.bf16, a.image_size, a.pipeline_model_parallel_size, a.reset_attention_mask = True, 28, 1, case["reset"]
    a.micro_batch_size, a.seq_length, a.create_attention_mask_in_dataloader = 2, 64, False
    dist.init_process_group("gloo", init_method=f"tcp://127.0.0.1:{port}", rank=rank, world_size=2)
    for mod in ("megatron.core.mpu", "megatron.core.parallel_state"):
        m = sys.modules[mod]
        m.get_tensor_model_parallel_rank = lambda: rank
        m.get_tensor_model_parallel_src_rank = lambda: 0
        m.get_tensor_model_parallel_group = lambda: None
        m.is_pipeline_first_stage = lambda: True
        m.is_pipeline_last_stage = lambda: True
    torch.Tensor.cuda = lambda self, *a_, **k_: self
    torch.cuda.current_device = lambda: "cpu"
    utils = importlib.import_module("long_vita_megatron.training.utils")
    batch = utils.get_batch_on_this_tp_rank(iter(tp_batch_data(case)) if rank == 0 else None)
    out = {k: v for k, v in batch.items()}
    out["actual_seq_len"] = utils.get_actual_seq_len() if case["reset"] else None
    torch.save(out, os.path.join(out_dir, f"rank{rank}.pt"))
    dist.barrier()
    dist.destroy_process_group()


def golden_tp_batch():
    out = {"cases": []}
    for case in TP_BATCH_CASES:
        ranks = _run_ranks(_tp_batch_worker, case)
        out["cases"].append(dict(case, ranks=ranks))
    torch.save(out, os.path.join(OUT, "tp_batch.pt"))


def golden_loss_func():
    out = {"cases": []}
    for case in LOSS_CASES:
        ranks = _run_ranks(_loss_worker, case)
        ins = [loss_case_inputs(case, r) for r in range(case["cp"])]
        out["cases"].append(dict(case, losses=[i[0] for i in ins], masks=[i[1] for i in ins], out=ranks))
    torch.save(out, os.path.join(OUT, "loss_func.pt"))


def _tree_map(p, f):
    if isinstance(p, dict):
        return {k: _tree_map(v, f) for k, v in p.items()}
    if isinstance(p, list):
        return [_tree_map(v, f) for v in p]
    return f(p)


HF_LV_SMALL = dict(llm=dict(num_layers=2, hidden=1024, heads=8, kv_groups=2, head_dim=128, ffn=2816, vocab=1024), vit_layers=2,
                   seq=600, image_at=5, new_tokens=6, weight_seed=77, input_seed=78)


def hf_long_vita_case(case=HF_LV_SMALL):
    """The seeded small LongVITAForCausalLM-layout checkpoint + request shared by the generator below and tests/test_hf_adaptor_gpu.py:
    -> (config dict like config_14B.json, state dict in bf16 with the HF class's names, input_ids [1, S], images [1, 3, 448, 448] bf16,
    image_indices [2, 1, 256])."""
    import json
    from oracle import llm as ollm, vit as ovit
    lc = ollm.LLMConfig(**case["llm"])
    vc = ovit.ViTConfig(num_layers=case["vit_layers"], llm_hidden=lc.hidden)
    lp = ollm.init_llm_params(lc, seed=case["weight_seed"])
    vp = ovit.init_vit_params(vc, seed=case["weight_seed"] + 1)
    sd = dict(ollm.to_hf_state_dict(lp, lc))
    sd.update(ovit.to_hf_state_dict(vp, vc, prefix="model.vision_model.", projector_prefix="model.vision_projection."))
    cfg_path = os.path.join(os.path.dirname(os.path.abspath(__file__)), "hf_long_vita_small_config.json")
    config = json.load(open(cfg_path))
    g = torch.Generator().manual_seed(case["input_seed"])
    ids = torch.randint(3, lc.vocab, (1, case["seq"]), generator=g)
    images = torch.randn(1, 3, 448, 448, generator=g).bfloat16()
    pos = torch.arange(case["image_at"], case["image_at"] + 256)
    idx = torch.stack([torch.zeros(1, 256, dtype=torch.long), pos[None]])
    return config, sd, ids, images, idx, lc, vc


def golden_hf_long_vita():
    """what `LongVITAForCausalLM.forward` / greedy `generate` compute (H/models/long_vita_qwen2_intern/
    modeling_long_vita.py:74-221, 238-327) on a small seeded checkpoint, fp32 on the CPU.  The class itself does not import against the
    transformers of this image (5.x: `LossKwargs` and the 4.48 decoder-layer call signature it was written for are gone), so its forward
    is composed here from the parts it is made of, each UNMODIFIED: the reference's own InternVisionModel and ResamplerProjector
    (imported from /root/reference as golden_hf_vit does), transformers' Qwen2ForCausalLM (the class LongVITAForCausalLM derives from),
    and the three lines of :137-147 that scatter the projected features into the embeddings."""
    import json

    import transformers
    timm = types.ModuleType("timm"); timm_m = types.ModuleType("timm.models"); timm_l = types.ModuleType("timm.models.layers")

    class DropPath(torch.nn.Identity):
        def __init__(self, *a, **k):
            super().__init__()

    timm_l.DropPath = DropPath
    sys.modules.update({"timm": timm, "timm.models": timm_m, "timm.models.layers": timm_l})
    for n in ["long_vita", "long_vita.models", "long_vita.models.long_vita_qwen2_intern"]:
        m = types.ModuleType(n)
        m.__path__ = [os.path.join(REF, *n.split("."))]
        sys.modules[n] = m
    civ = importlib.import_module("long_vita.models.long_vita_qwen2_intern.configuration_intern_vit")
    miv = importlib.import_module("long_vita.models.long_vita_qwen2_intern.modeling_intern_vit")
    rp = importlib.import_module("long_vita.models.long_vita_qwen2_intern.resampler_projector")
    case = HF_LV_SMALL
    config, sd, ids, images, idx, lc, vc = hf_long_vita_case(case)
    sd = {k: v.float() for k, v in sd.items()}
    vcfg = dict(config["visual"]); vcfg["use_flash_attn"] = False
    hcfg = civ.InternVisionConfig(**vcfg)
    vision_model = miv.InternVisionModel(hcfg).eval().float()
    projection = rp.ResamplerProjector(types.SimpleNamespace(hidden_size=lc.hidden), hcfg).eval().float()
    vision_model.load_state_dict({k[len("model.vision_model."):]: v for k, v in sd.items() if k.startswith("model.vision_model.")})
    projection.load_state_dict({k[len("model.vision_projection."):]: v for k, v in sd.items() if k.startswith("model.vision_projection.")})
    qcfg = transformers.Qwen2Config(vocab_size=lc.vocab, hidden_size=lc.hidden, intermediate_size=lc.ffn, num_hidden_layers=lc.num_layers,
                                    num_attention_heads=lc.heads, num_key_value_heads=lc.kv_groups, rms_norm_eps=lc.eps, rope_theta=lc.rope_theta,
                                    max_position_embeddings=4096, tie_word_embeddings=False, attention_dropout=0.0)
    qcfg._attn_implementation = "eager"
    llm = transformers.Qwen2ForCausalLM(qcfg).eval().float()
    missing = llm.load_state_dict({k: v for k, v in sd.items() if not k.startswith(("model.vision_model.", "model.vision_projection."))}, strict=False)
    assert not missing.unexpected_keys and all("rotary" in k or "inv_freq" in k for k in missing.missing_keys), missing

    def forward(tokens):
        with torch.no_grad():
            image_embeds = vision_model(images.float()).last_hidden_state              # :91
            image_embeds = projection(image_embeds[:, 1:, :])                         # :97-98
            inputs_embeds = llm.model.embed_tokens(tokens)                            # :137
            inputs_embeds = inputs_embeds.clone()                                     # :142-147
            indices_b, indices_s = idx.unbind(dim=0)
            inputs_embeds[indices_b.view(-1), indices_s.view(-1)] = image_embeds.view(-1, image_embeds.shape[-1])
            return llm(inputs_embeds=inputs_embeds).logits, image_embeds, inputs_embeds

    logits, image_embeds, embeds = forward(ids)
    tokens, gaps = ids.clone(), []
    for _ in range(case["new_tokens"]):                                               # greedy search, no cache: the whole row again
        last = forward(tokens)[0][0, -1]
        top = torch.topk(last, 2).values
        gaps.append(float(top[0] - top[1]))
        tokens = torch.cat([tokens, last.argmax().view(1, 1)], dim=1)
    torch.save(dict(case=case, logits_rows=logits[0, ::7].clone(), logits_last=logits[0, -1].clone(), image_embeds_sub=image_embeds[0, ::8, ::8].clone(),
                    embeds_sub=embeds[0, ::5, ::16].clone(), generated=tokens[0, ids.shape[1]:].clone(), top2_gaps=torch.tensor(gaps),
                    logits_rms=float(logits.pow(2).mean().sqrt())), os.path.join(OUT, "hf_long_vita.pt"))


def main():
    os.makedirs(OUT, exist_ok=True)
    _install_stubs()
    only = set(sys.argv[1:])                              # python -m oracle.make_golden [fixture ...]
    if only == {"_patch_manager_inproc"}:
        return _patch_manager_inproc()
    for name, fn in [("cp_slice", golden_cp_slice), ("rope_rmsnorm", golden_rope_rmsnorm),
                     ("embedding_scatter", golden_embedding_scatter), ("masked_linear", golden_masked_linear),
                     ("hf_vit", golden_hf_vit), ("image_processor", golden_image_processor),
                     ("external_inputs", golden_external_inputs), ("decode_loop", golden_decode_loop), ("loss_func", golden_loss_func), ("unfused_attention", golden_unfused_attention),
                     ("unfused_attention_bf16", golden_unfused_attention_bf16),
                     ("converters", golden_converters), ("sampling", golden_sampling), ("patch_manager", golden_patch_manager),
                     ("adaptor_targets", golden_adaptor_targets), ("packed_positions", golden_packed_positions),
                     ("ckpt_scripts", golden_ckpt_scripts), ("tp_batch", golden_tp_batch), ("hf_long_vita", golden_hf_long_vita)]:
        if only and name not in only:
            continue
        fn()
        print(name, "ok")
    # r05: the reference's composite classes over plain-torch leaves (oracle/make_golden_composites.py).  Last: they re-bind names on
    # the stub modules and re-import reference modules.
    from . import make_golden_composites as comp
    for name, fn in [("gptvl_forward", lambda: comp.golden_gptvl_forward(OUT, STATE)),
                     ("transformer_block", lambda: comp.golden_transformer_block(OUT, STATE)),
                     ("intern_vit_forward", lambda: comp.golden_intern_vit_forward(OUT, STATE, cpu_as_cuda)),
                     ("vision_model", lambda: comp.golden_vision_model(OUT, STATE)),
                     ("forward_step", lambda: comp.golden_forward_step(OUT, STATE))]:
        if only and name not in only:
            continue
        fn()
        print(name, "ok")


if __name__ == "__main__":
    main()
