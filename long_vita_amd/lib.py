"""ctypes binding of libvita_hip.so — the C-ABI boundary declared in include/vita_hip.h.

There is deliberately NO fallback: if the shared library is missing or a symbol cannot be
resolved the import of the product path fails loudly (``VitaLibraryError``).  Nothing in this
package may import ``oracle/`` (the CPU checker).

Error mapping mirrors the reference's Python exception types (SURVEY.md §8b "Errors"):
  VITA_ERR_INVALID_ARG -> ValueError, VITA_ERR_UNSUPPORTED -> RuntimeError (the reference raises
  RuntimeError on weight-shape mismatch, M/core/tensor_parallel/layers.py:849-853),
  VITA_ERR_LAUNCH -> RuntimeError.
"""
from __future__ import annotations

import ctypes as C
import os
import subprocess
import threading

_HERE = os.path.dirname(os.path.abspath(__file__))
CSRC_DIR = os.path.join(_HERE, "csrc")
LIB_PATH = os.path.join(CSRC_DIR, "libvita_hip.so")
if os.environ.get("VITA_HIP_LIB"):                    # developer A / B switch: another build of the SAME library (tools/microbench.py, same box)
    LIB_PATH = os.path.abspath(os.environ["VITA_HIP_LIB"])
HEADER_PATH = os.path.join(os.path.dirname(_HERE), "include", "vita_hip.h")

ABI_VERSION = 18
VITA_OK = 0
VITA_ERR_INVALID_ARG = -1
VITA_ERR_UNSUPPORTED = -2
VITA_ERR_LAUNCH = -3

(EPI_NONE, EPI_BIAS, EPI_BIAS_GELU, EPI_RESIDUAL, EPI_BIAS_SCALE_RES, EPI_SWIGLU, EPI_BIAS2_GELU_TANH, EPI_BIAS2_RES,
 EPI_BIAS2_GELU) = range(9)


class VitaLibraryError(ImportError):
    """libvita_hip.so is missing / unbuildable / lacks a symbol."""


class AttnParams(C.Structure):
    """Mirror of ``vita_attn_params`` (include/vita_hip.h)."""

    _fields_ = [
        ("q", C.c_void_p), ("q_batch_stride", C.c_int64), ("q_row_stride", C.c_int64), ("q_head_stride", C.c_int64), ("q_group_stride", C.c_int64),
        ("k", C.c_void_p), ("k_batch_stride", C.c_int64), ("k_row_stride", C.c_int64), ("k_head_stride", C.c_int64),
        ("v", C.c_void_p), ("v_batch_stride", C.c_int64), ("v_row_stride", C.c_int64), ("v_head_stride", C.c_int64),
        ("o", C.c_void_p), ("o_batch_stride", C.c_int64), ("o_row_stride", C.c_int64), ("o_head_stride", C.c_int64), ("o_group_stride", C.c_int64),
        ("lse", C.c_void_p),
        ("batch", C.c_int), ("n_q_heads", C.c_int), ("n_kv_heads", C.c_int), ("head_dim", C.c_int),
        ("chunk_len", C.c_int64), ("q_valid", C.c_int64), ("kv_valid", C.c_int64),
        ("n_q_chunks", C.c_int), ("n_kv_chunks", C.c_int),
        ("q_chunk_gid", C.POINTER(C.c_int32)),
        ("kv_chunk_gid", C.POINTER(C.c_int32)),
        ("kv_chunk_row", C.POINTER(C.c_int64)),
        ("causal", C.c_int),
        ("softmax_scale", C.c_float),
        ("q_seg_start", C.c_void_p),
    ]


class AttnBwdParams(C.Structure):
    """Mirror of ``vita_attn_bwd_params`` (include/vita_hip.h)."""

    _fields_ = [
        ("q", C.c_void_p), ("q_row_stride", C.c_int64), ("q_head_stride", C.c_int64), ("q_group_stride", C.c_int64),
        ("k", C.c_void_p), ("k_row_stride", C.c_int64), ("k_head_stride", C.c_int64),
        ("v", C.c_void_p), ("v_row_stride", C.c_int64), ("v_head_stride", C.c_int64),
        ("d_o", C.c_void_p), ("do_row_stride", C.c_int64), ("do_head_stride", C.c_int64),
        ("lse", C.c_void_p),
        ("delta", C.c_void_p),
        ("dq", C.c_void_p), ("dq_row_stride", C.c_int64), ("dq_head_stride", C.c_int64), ("dq_group_stride", C.c_int64),
        ("dk", C.c_void_p), ("dk_row_stride", C.c_int64), ("dk_head_stride", C.c_int64),
        ("dv", C.c_void_p), ("dv_row_stride", C.c_int64), ("dv_head_stride", C.c_int64),
        ("n_q_heads", C.c_int), ("n_kv_heads", C.c_int), ("head_dim", C.c_int),
        ("chunk_len", C.c_int64),
        ("n_q_chunks", C.c_int), ("n_kv_chunks", C.c_int),
        ("q_chunk_gid", C.POINTER(C.c_int32)),
        ("kv_chunk_gid", C.POINTER(C.c_int32)),
        ("kv_chunk_row", C.POINTER(C.c_int64)),
        ("softmax_scale", C.c_float),
        ("q_seg_start", C.c_void_p), ("k_seg_end", C.c_void_p),
    ]


class CpAttnParams(C.Structure):
    """Mirror of ``vita_cp_attn_params`` (include/vita_hip.h)."""

    _fields_ = [
        ("q", C.c_void_p), ("q_row_stride", C.c_int64), ("q_head_stride", C.c_int64), ("q_group_stride", C.c_int64),
        ("kv_packed", C.c_void_p),
        ("out", C.c_void_p), ("out_row_stride", C.c_int64), ("out_head_stride", C.c_int64),
        ("lse", C.c_void_p),
        ("s_local", C.c_int64),
        ("n_q_heads", C.c_int), ("n_kv_heads", C.c_int), ("head_dim", C.c_int), ("n_split", C.c_int),
        ("softmax_scale", C.c_float),
        ("workspace", C.c_void_p), ("workspace_bytes", C.c_size_t),
        ("dkv_workspace", C.c_void_p),
        ("scratch", C.c_void_p), ("scratch_bytes", C.c_size_t),
    ]


class DecodeLayerParams(C.Structure):
    """Mirror of ``vita_decode_layer_params`` (include/vita_hip.h)."""

    _fields_ = [
        ("ln1", C.c_void_p), ("qkv_w", C.c_void_p), ("qkv_b", C.c_void_p), ("o_w", C.c_void_p), ("ln2", C.c_void_p),
        ("fc1_w", C.c_void_p), ("fc2_w", C.c_void_p),
        ("hidden", C.c_int), ("heads", C.c_int), ("kv_groups", C.c_int), ("head_dim", C.c_int), ("ffn", C.c_int),
        ("eps", C.c_float), ("softmax_scale", C.c_float),
        ("h", C.c_void_p),
        ("cos", C.c_void_p), ("sin", C.c_void_p),
        ("k_cache", C.c_void_p), ("v_cache", C.c_void_p),
        ("kv_row_stride", C.c_int64), ("kv_group_stride", C.c_int64),
        ("capacity", C.c_int),
        ("append_row", C.c_int),
        ("len", C.c_int),
        ("n_splits", C.c_int),
        ("qkv", C.c_void_p), ("ctx", C.c_void_p), ("act", C.c_void_p),
        ("part_m", C.c_void_p), ("part_l", C.c_void_p), ("part_o", C.c_void_p),
        ("msg", C.c_void_p),
        ("gathered", C.c_void_p),
        ("n_ranks", C.c_int),
    ]


_p, _i, _l, _f = C.c_void_p, C.c_int, C.c_int64, C.c_float

# name -> (restype, argtypes); must list EVERY function declared in include/vita_hip.h
PROTOTYPES = {
    "vita_abi_version": (_i, []),
    "vita_error_string": (C.c_char_p, [_i]),
    "vita_rmsnorm_fwd": (_i, [_p, _p, _p, _p, _l, _i, _f, _p]),
    "vita_layernorm_fwd": (_i, [_p, _p, _p, _p, _l, _i, _f, _p]),
    "vita_logit_postprocess": (_i, [_p, _l, _l, _l, _f, _f, _p]),
    "vita_logit_postprocess_bwd": (_i, [_p, _l, _p, _l, _l, _l, _f, _f, _p]),
    "vita_rope_table": (_i, [_p, _p, _p, _p, _l, _i, _p]),
    "vita_rope_cos_sin": (_i, [_p, _l, _p, _p, _l, _i, _p]),
    "vita_rope_apply": (_i, [_p, _l, _i, _i, _l, _l, _p, _p, _i, _p]),
    "vita_rope_qkv_fwd": (_i, [_p, _l, _i, _i, _i, _p, _p, _p, _i, _p]),
    "vita_row_gather": (_i, [_p, _l, _p, _p, _l, _i, _i, _p, _p]),
    "vita_row_scatter": (_i, [_p, _l, _p, _p, _l, _p, _l, _i, _i, _p, _p]),
    "vita_mask_to_index": (_i, [_p, _l, _p, _p, _p]),
    "vita_cp_index_remap": (_i, [_p, _l, _l, _i, _i, _p, _p, _p]),
    "vita_rows_any": (_i, [_p, _l, _i, _p, _p]),
    "vita_index_inverse": (_i, [_p, _l, _p, _p]),
    "vita_cp_src_tgt": (_i, [_p, _l, _i, _p, _p, _p, _p, _p, _p, _p, _p]),
    "vita_gemm_bf16": (_i, [_p, _l, _p, _l, _p, _l, _l, _l, _l, _i, _p, _p, _p, _l, _p]),
    "vita_gemm_bf16_tn": (_i, [_p, _l, _p, _l, _p, _l, _l, _l, _l, _p]),
    "vita_gemm_bf16_nn": (_i, [_p, _l, _p, _l, _p, _l, _l, _l, _l, _p]),
    "vita_gemm_tn_splitk_workspace_bytes": (C.c_size_t, [_l, _l, _i]),
    "vita_gemm_bf16_tn_splitk": (_i, [_p, _l, _p, _l, _p, _l, _l, _l, _l, _i, _p, _p]),
    "vita_colsum_bf16": (_i, [_p, _l, _p, _l, _i, _p]),
    "vita_colsum_workspace_bytes": (C.c_size_t, [_l, _i]),
    "vita_colsum_bf16_ordered": (_i, [_p, _l, _p, _l, _i, _p, _p]),
    "vita_gemm_skinny_bf16": (_i, [_p, _l, _p, _l, _p, _l, _i, _l, _l, _i, _p]),
    "vita_flash_attn_fwd": (_i, [C.POINTER(AttnParams), _p]),
    "vita_patchify14": (_i, [_p, _p, _l, _i, _i, _i, _p]),
    "vita_vit_assemble": (_i, [_p, _p, _p, _p, _l, _i, _i, _i, _p]),
    "vita_pixel_shuffle_ln": (_i, [_p, _p, _p, _p, _l, _i, _i, _i, _f, _p]),
    "vita_patchify14_ex": (_i, [_p, _p, _l, _i, _i, _i, _i, _p]),
    "vita_vit_assemble_ex": (_i, [_p, _p, _p, _p, _l, _i, _i, _i, _i, _i, _p]),
    "vita_vit_assemble_bwd": (_i, [_p, _p, _l, _i, _i, _i, _p]),
    "vita_pixel_shuffle_ln_ex": (_i, [_p, _p, _p, _p, _l, _i, _i, _i, _f, _i, _l, _l, _p]),
    "vita_pixel_shuffle_ln_bwd": (_i, [_p, _p, _p, _p, _p, _p, _l, _i, _i, _i, _f, _i, _l, _l, _p]),
    # backward
    "vita_rope_qkv_bwd": (_i, [_p, _l, _i, _i, _i, _p, _p, _p]),
    "vita_transpose_bf16": (_i, [_p, _l, _p, _l, _l, _l, _p]),
    "vita_rmsnorm_bwd": (_i, [_p, _p, _p, _p, _p, _p, _l, _i, _f, _p]),
    "vita_swiglu_fwd": (_i, [_p, _p, _l, _i, _p]),
    "vita_swiglu_bwd": (_i, [_p, _p, _p, _l, _i, _p]),
    "vita_gelu_bwd": (_i, [_p, _p, _p, _l, _p]),
    "vita_gelu_tanh_bwd": (_i, [_p, _p, _p, _l, _p]),
    "vita_layernorm_param_grad": (_i, [_p, _p, _p, _p, _l, _i, _f, _i, _p]),
    "vita_layernorm_bwd": (_i, [_p, _p, _p, _p, _p, _p, _l, _i, _f, _p]),
    "vita_gelu_fwd": (_i, [_p, _p, _l, _i, _p]),
    "vita_bias_scale_res_fwd": (_i, [_p, _p, _p, _p, _p, _l, _i, _p]),
    "vita_bias_scale_res_bwd": (_i, [_p, _p, _p, _p, _p, _p, _p, _l, _i, _p]),
    "vita_ce_loss": (_i, [_p, _l, _p, _p, _p, _l, _p, _l, _i, _p, _p]),
    "vita_ce_loss_f32": (_i, [_p, _l, _p, _p, _p, _l, _p, _l, _i, _p, _p]),
    "vita_ce_vp_stats": (_i, [_p, _i, _l, _p, _l, _p, _l, _i, _p]),
    "vita_ce_vp_finish": (_i, [_p, _i, _l, _p, _p, _p]),
    "vita_ce_vp_grad": (_i, [_p, _i, _l, _p, _l, _p, _p, _p, _l, _l, _i, _p]),
    "vita_row_scatter_add_f32": (_i, [_p, _p, _p, _l, _l, _i, _p, _p]),
    "vita_attn_delta": (_i, [_p, _p, _p, _l, _i, _i, _l, _l, _l, _l, _p]),
    "vita_flash_attn_bwd": (_i, [C.POINTER(AttnBwdParams), _p]),
    "vita_flash_attn_bwd_parts": (_i, [C.POINTER(AttnBwdParams), _i, _p]),
    "vita_attn_merge": (_i, [_p, _l, _l, _p, _p, _l, _l, _p, _l, _i, _i, _p]),
    "vita_gemv_bf16": (_i, [_p, _p, _l, _p, _l, _l, _i, _p, _p, _p]),
    "vita_decode_attn_partial": (_i, [_p, _l, _l, _p, _p, _l, _l, _i, _p, _i, _i, _i, _i, _f, _p, _p, _p, _p]),
    "vita_add_bf16": (_i, [_p, _p, _p, _l, _p]),
    "vita_decode_layer_attn": (_i, [C.POINTER(DecodeLayerParams), _p]),
    "vita_decode_layer_mlp": (_i, [C.POINTER(DecodeLayerParams), _p]),
    "vita_frames_resize_norm": (_i, [_p, _l, _i, _i, _i, _i, C.POINTER(C.c_int), _i, _i, _i, _i, _i, _i, _i, _p, _p, _i, _p, _p, _i,
                                      C.POINTER(C.c_float), C.POINTER(C.c_float), _p, _p, _p, _p]),
    "vita_cp_unique_id": (_i, [_p]),
    "vita_cp_init": (_i, [C.POINTER(_p), _i, _i, _p]),
    "vita_cp_destroy": (_i, [_p]),
    "vita_cp_attn_workspace_bytes": (C.c_size_t, [_i, _l, _i, _i]),
    "vita_cp_attn_scratch_bytes": (C.c_size_t, [_l, _i, _i, _i]),
    "vita_cp_attn_fwd": (_i, [_p, C.POINTER(CpAttnParams), _p]),
    "vita_cp_attn_bwd": (_i, [_p, C.POINTER(CpAttnParams), _p, _p, _p, _p, _p, _p]),
    "vita_decode_attn_merge": (_i, [_p, _p, _p, _i, _l, _l, _i, _i, _p, _p, _p, _p, _p]),
}

_lock = threading.Lock()
_lib = None


def build(force: bool = False) -> str:
    """Compile csrc/*.hip for gfx950 into csrc/libvita_hip.so (hipcc cross-compiles on CPU)."""
    cmd = ["make", "-C", CSRC_DIR, "-j8"]
    if force:
        subprocess.run(["make", "-C", CSRC_DIR, "clean"], check=True, capture_output=True)
    r = subprocess.run(cmd, capture_output=True, text=True)
    if r.returncode != 0:
        raise VitaLibraryError(f"building libvita_hip.so failed:\n{r.stdout}\n{r.stderr}")
    return LIB_PATH


def load(allow_build: bool = True):
    """Load the library once; torch must already be imported so that its bundled HIP runtime
    (SONAME libamdhip64.so.7) is the one the kernels' stream handles belong to."""
    global _lib
    with _lock:
        if _lib is not None:
            return _lib
        import torch  # noqa: F401  (loads libamdhip64 first)

        if not os.path.exists(LIB_PATH):
            if not allow_build:
                raise VitaLibraryError(f"{LIB_PATH} not found (run `python -c 'import __graft_entry__ as g; g.build()'`)")
            build()
        try:
            lib = C.CDLL(LIB_PATH)
        except OSError as e:  # pragma: no cover
            raise VitaLibraryError(f"cannot load {LIB_PATH}: {e}") from e
        for name, (res, args) in PROTOTYPES.items():
            try:
                fn = getattr(lib, name)
            except AttributeError as e:
                raise VitaLibraryError(f"{LIB_PATH} does not export {name}") from e
            fn.restype = res
            fn.argtypes = args
        if lib.vita_abi_version() != ABI_VERSION:
            raise VitaLibraryError("libvita_hip.so ABI version mismatch")
        _lib = lib
        return _lib


def check(code: int, what: str = "") -> None:
    """Convert a C return code into the reference's exception types."""
    if code == VITA_OK:
        return
    msg = load().vita_error_string(code).decode()
    text = f"{what}: {msg}" if what else msg
    if code == VITA_ERR_INVALID_ARG:
        raise ValueError(text)
    raise RuntimeError(text)
